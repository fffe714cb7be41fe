#!/bin/bash
# tools/ntt_diag.sh -- builds two DIAGNOSTIC variants of the library (wrong results by construction; never shipped):
#   libmi355zk_nostages.so  the NTT passes load, bit-reverse into LDS and store, but run no butterflies   -> the I/O share of a pass
#   libmi355zk_noio.so      the passes run all their butterflies on synthetic data and write nothing      -> the arithmetic share
# then (on the GPU box) times them with tools/ntt_time_one.py through MZK_LIB_PATH.   bash tools/ntt_diag.sh build | run
set -e
cd "$(dirname "$0")/../mpc-jellyfish_amd/csrc"
if [ "$1" = build ]; then
  F="-O3 -std=c++17 --offload-arch=gfx950 -fPIC -fvisibility=hidden -Wno-unused-result -ffp-contract=off"
  for v in NOSTAGES NOIO; do
    lc=$(echo $v | tr A-Z a-z)
    /opt/rocm/bin/hipcc $F -DMZK_NTT_DIAG_$v -c -o build/ntt_$lc.o ntt.hip
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libmi355zk_$lc.so build/mzk.o build/ntt_$lc.o build/msm.o build/plonk.o build/poly.o
  done
else
  cd ../..
  for v in "" _nostages _noio; do
    echo "== libmi355zk$v.so"
    MZK_LIB_PATH=$PWD/mpc-jellyfish_amd/libmi355zk$v.so python3 tools/ntt_time_one.py 22 20
  done
fi
