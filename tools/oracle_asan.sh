#!/bin/bash
# CPU sanitizer run of the oracle (test infrastructure): AddressSanitizer + UBSan build of oracle/ocean_oracle.c driven
# through every mode, FFT stage variant and team size.  (GPU sanitizers are not available on the pool.)
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
gcc -O1 -g -std=gnu11 -fPIC -fopenmp -mavx -ffp-contract=off -fsanitize=address,undefined -fno-omit-frame-pointer -shared \
    -o /tmp/libocean_oracle_asan.so $R/oracle/ocean_oracle.c -lm -ldl
cat > /tmp/oracle_asan_run.py <<PY
import sys
sys.path.insert(0, "$R")
from oracle import oracle as O
O._LIB_PATH = "/tmp/libocean_oracle_asan.so"
O.build = lambda force=False: O._LIB_PATH
for n in (16, 64, 128):
    o = O.Oracle(n, lam=-1.5); o.prepare(seed=3)
    for mode in (0, 1, 2, 3):
        for fft in (O.FFT_F32, O.FFT_F64, O.FFT_F32_TEAM):
            for th in (2, 8, 5):
                O.lib().oracle_set_num_threads(th)
                o.compute_waves(1.0, mode=mode, fft=fft)
    o.use_pocketfft(2); o.compute_waves(0.5, mode=3, fft=O.FFT_EXTERNAL)
    o2 = O.Oracle(n, dispersion=(1, 20.0)); o2.prepare(xi=O.gauss_xi_numpy(1, n)); o2.compute_waves(2.0)
print("oracle sanitizer run: clean")
PY
LD_PRELOAD=$(gcc -print-file-name=libasan.so) ASAN_OPTIONS=detect_leaks=0 python /tmp/oracle_asan_run.py
