#!/bin/sh
# host build of the shared per-element math header (test infrastructure only)
cd "$(dirname "$0")" && g++ -O2 -shared -fPIC -ffp-contract=off -Wno-unknown-pragmas -o libhostmath.so hostmath.cpp
