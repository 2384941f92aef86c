#!/bin/bash
# AddressSanitizer + UBSan + LeakSanitizer over the CPU side of the product: the drop-in host layer (box2d-mt_amd/host) on
# top of the C oracle's ABI shim, driven by every harness scene (recording listener, user filter, the scripted life-cycle
# scene) and - when /root/reference is present - by the reference's own Testbed scene headers. GPU sanitizers are not
# available on the pool; this covers the host logic the device path shares (b2World / b2Body / b2Fixture, callbacks,
# per-body caches touched from user range tasks). Output: gpurun_out/asan/{harness,testbed}.log; exit code 1 on any report.
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
O=$R/gpurun_out/asan
mkdir -p $O && cd $O
SAN="-fsanitize=address,undefined -fno-omit-frame-pointer -g -O1"
for f in b2o_step b2o_collide b2o_joint b2o_toi b2o_abi_shim; do gcc $SAN -std=c11 -ffp-contract=off -fPIC -I$R/include -c $R/oracle/$f.c -o $f.o; done
g++ $SAN -std=c++17 -ffp-contract=off -w -DB2H_BACKEND_AMD -I$R/box2d-mt_amd/host -I$R/include -I$R/box2d-mt_amd/harness -o harness_asan \
    $R/tools/sanitize/harness_main.cpp $R/box2d-mt_amd/harness/harness.cpp $R/box2d-mt_amd/host/src/*.cpp b2o_*.o -lpthread -lm
ASAN_OPTIONS=detect_leaks=1 ./harness_asan > harness.log 2>&1 || true
bad=$(grep -c -E "ERROR|runtime error" harness.log || true)
if [ -d /root/reference/Testbed ]; then
  g++ $SAN -std=c++17 -ffp-contract=off -w -I$R/box2d-mt_amd/host -I$R/include -I/root/reference -I$R/tests/testbed -o testbed_asan \
      $R/tools/sanitize/testbed_main.cpp $R/tests/testbed/scenes_main.cpp $R/box2d-mt_amd/host/src/*.cpp b2o_*.o -lpthread -lm
  ASAN_OPTIONS=detect_leaks=1 ./testbed_asan > testbed.log 2>&1 || true
  bad=$((bad + $(grep -c -E "ERROR|runtime error" testbed.log || true)))
fi
echo "sanitizer reports: $bad"; [ "$bad" = "0" ]
