#!/bin/bash
# AddressSanitizer + UBSan over the CPU-side code (the oracle and the host mirror) with the CPU test suite.
# GPU sanitizers are not available on this pool; the device code is covered by the parity tests instead.
# Run from the repo root after `python -m zinc_amd.build` (the mirror links against libzip_hip.so).
set -e
TMP=$(mktemp -d)
PRE="$(gcc -print-file-name=libasan.so):$(gcc -print-file-name=libubsan.so)"
SAN="-fsanitize=address,undefined -fno-omit-frame-pointer -O1 -g"
gcc $SAN -std=c11 -fPIC -fopenmp -shared -o $TMP/libzip_oracle.so oracle/zip_oracle.c
g++ $SAN -std=c++17 -fPIC -shared -Iinclude -Izinc_amd/host -o $TMP/libzinc_zip.so zinc_amd/host/c_api.cpp zinc_amd/host/zinc_zip.cpp \
    -Lzinc_amd/lib -lzip_hip -Wl,-rpath,$PWD/zinc_amd/lib
cp oracle/_build/libzip_oracle.so $TMP/oracle.orig; cp zinc_amd/lib/libzinc_zip.so $TMP/mirror.orig
restore() { cp $TMP/oracle.orig oracle/_build/libzip_oracle.so; cp $TMP/mirror.orig zinc_amd/lib/libzinc_zip.so; touch oracle/_build/libzip_oracle.so zinc_amd/lib/libzinc_zip.so; }
trap restore EXIT
cp $TMP/libzip_oracle.so oracle/_build/libzip_oracle.so; cp $TMP/libzinc_zip.so zinc_amd/lib/libzinc_zip.so
touch oracle/_build/libzip_oracle.so zinc_amd/lib/libzinc_zip.so
ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 LD_PRELOAD=$PRE \
    python -m pytest tests/test_oracle_kats.py tests/test_oracle_protocol.py tests/test_oracle_sumcheck.py tests/test_oracle_spartan.py \
    tests/test_host_mirror.py -x -q -p no:cacheprovider
