# us per transform across sizes (planner anomalies show as jumps in ns per element-layer)
for n in 13 14 15 16 17 18 19 20 21 22 23 24 25 26; do
  c=$(( n <= 20 ? 256 : (n <= 24 ? 32 : 8) ))
  python tools/cfft_time.py --cols $c --log $n --reps 40 | tail -1
done
