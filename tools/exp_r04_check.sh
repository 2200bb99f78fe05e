#!/bin/bash
# digests of every plan exp_r04_cfft.sh times (equal within a size = equal results)
E=$PWD/tstwo_amd/libtstwo_hip_exp.so
python tools/plan_digest.py --log 22
TSTWO_HIP_LIB=$E TSTWO_CFFT_KB=13 TSTWO_CFFT_KA=9 TSTWO_CFFT_LOGTA=15 python tools/plan_digest.py --log 22
TSTWO_HIP_LIB=$E TSTWO_CFFT_KB=14 TSTWO_CFFT_KA=8 TSTWO_CFFT_LOGTA=15 python tools/plan_digest.py --log 22
TSTWO_HIP_LIB=$PWD/build/exp/a_sb.so python tools/plan_digest.py --log 22
python tools/plan_digest.py --log 23
TSTWO_HIP_LIB=$E TSTWO_CFFT_KB=14 TSTWO_CFFT_KA=9 TSTWO_CFFT_LOGTA=15 python tools/plan_digest.py --log 23
TSTWO_HIP_LIB=$E TSTWO_CFFT_KB=13 TSTWO_CFFT_KA=10 TSTWO_CFFT_LOGTA=15 python tools/plan_digest.py --log 23
python tools/plan_digest.py --log 24
TSTWO_HIP_LIB=$E TSTWO_CFFT_KB=13 TSTWO_CFFT_KA=9 TSTWO_CFFT_LOGTA=14 python tools/plan_digest.py --log 24
