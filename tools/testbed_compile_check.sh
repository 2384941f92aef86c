#!/bin/bash
# Syntax-checks every scene header of the reference's Testbed (Testbed/Tests/*.h, included where they lie, unmodified) against
# the drop-in Box2D headers of box2d-mt_amd/host with the headless Test stub (tests/testbed/headless_test.h).
# Usage: tools/testbed_compile_check.sh [-v]   (prints "<n> of <m> compile"; -v also prints the first errors of each failure)
R=$(cd "$(dirname "$0")/.." && pwd)
REF=${REF:-/root/reference}
ok=0; n=0; failed=""
for h in $REF/Testbed/Tests/*.h; do
  b=$(basename $h .h)
  n=$((n+1))
  printf '#include "headless_test.h"\nDebugDraw g_debugDraw; Camera g_camera;\n#include "Testbed/Tests/%s.h"\n' $b > /tmp/tb_$b.cpp
  if g++ -std=c++17 -fsyntax-only -w -I$R/box2d-mt_amd/host -I$R/include -I$REF -I$R/tests/testbed /tmp/tb_$b.cpp 2>/tmp/tb_$b.err; then ok=$((ok+1)); else failed="$failed $b"; fi
  rm -f /tmp/tb_$b.cpp
done
echo "$ok of $n compile"
for b in $failed; do
  echo "FAIL $b"
  if [ "$1" = "-v" ]; then grep -m4 "error" /tmp/tb_$b.err | cut -c1-220; fi
done
