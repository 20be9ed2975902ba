#!/bin/bash
# csrc/libm_glibc.h (host build) against the C library on every float32 argument and 2^32 float64 patterns (8 cores: 35 s)
set -e
g++ -O2 -fopenmp -ffp-contract=off -std=c++17 -I picasso_amd/csrc tests/native/libm_glibc_exhaustive.cpp -o /tmp/libm_glibc_exhaustive
/tmp/libm_glibc_exhaustive
