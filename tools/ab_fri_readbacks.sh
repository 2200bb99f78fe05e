# A/B on one box: FRI commit with one download_many at the end against one read-back per piece
python tools/bench_fri_sizes.py 2>&1 | grep "tstwo_fri_commit_layers" | sed 's/^/many: /'
TSTWO_FRI_SEPARATE_READBACKS=1 python tools/bench_fri_sizes.py 2>&1 | grep "tstwo_fri_commit_layers" | sed 's/^/sep:  /'
python tools/bench_fri_sizes.py 2>&1 | grep "tstwo_fri_commit_layers" | sed 's/^/many: /'
