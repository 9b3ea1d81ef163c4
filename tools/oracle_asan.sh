# CPU-only: the oracle built with AddressSanitizer + UBSan, then the "not gpu" test suite run against it (sanitizers are not available on the GPU pool)
set -e
cd "$(dirname "$0")/.."
mkdir -p oracle/_build/asan
gcc -O1 -g -fPIC -fopenmp -std=gnu11 -fsanitize=address,undefined -fno-sanitize-recover=undefined -shared -o oracle/_build/asan/liboracle.so oracle/sfgwas_oracle.c -lquadmath -lm
LD_PRELOAD="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libubsan.so)" ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 UBSAN_OPTIONS=print_stacktrace=1 \
  SFG_ORACLE_SO=$PWD/oracle/_build/asan/liboracle.so OMP_NUM_THREADS=4 python -m pytest tests -x -q -m "not gpu" -p no:cacheprovider "$@"
