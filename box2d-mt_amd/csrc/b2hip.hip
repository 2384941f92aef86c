// b2hip.hip - libb2hip.so: host bookkeeping + step orchestration + the C ABI declared in include/b2hip.h.
//
// The world lives in HBM (b2d_world.h). The host keeps (a) the immutable per-body / per-fixture
// parameters it was given, (b) a mirror of the dynamic body state refreshed by the mandatory
// read-back at the end of each step, and (c) the deterministic proxy-id allocator that reproduces
// the ids the reference's dynamic tree would hand out (they define every deterministic ordering,
// b2ContactManager.cpp:64-92). All physics runs in HIP kernels; there is no CPU fallback.
#include <hip/hip_runtime.h>

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <dlfcn.h>
#include <rccl/rccl.h> // (types only: the library is opened with dlopen by b2hip_shard_connect)
#include <atomic>
#include <chrono>
#include <thread>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/b2hip.h"
#include "b2d_kernels_toi_chains.h"
#include "b2d_kernels_toi_domains.h"
// (The three earlier resident large-island solvers of rounds 1 - 2 - grid barrier per colour, polled body rows, pushed
// mailboxes - lived on as a test build until round 3 and were removed in round 4: k_solve_blocks / k_blocks_sweep are
// cross-checked against the launch-per-colour kernels, tests/test_gpu_parity.py. What is left of them compiles out.)
#include "b2d_handover.h"
#define B2HIP_HAVE_VALIDATION_SOLVERS 0
#include "b2d_kernels_solve_blocks.h"
#include "b2d_kernels_sweep_end.h"
#include "b2d_kernels_edit.h"
#include "b2d_kernels_shard.h"
#include "b2d_kernels_spatial.h"
#include "b2d_scan.h"
#include "b2d_shape_geom.h"

// ---- the host side, by concern (ONE translation unit: the kernels live in headers, and hipcc compiles the device code of every
// unit that includes them - seven units would be seven device compiles and seven copies of the kernels; the files below are
// consecutive pieces of what used to be one 6 400-line file and share its scope) ------------------------------------------
#include "b2hip_host_world.h"
#include "b2hip_host_phases.h"

extern "C"
{
#include "b2hip_api_world.h"
#include "b2hip_api_step.h"
#include "b2hip_api_snapshot.h"
#include "b2hip_api_sharding.h"
#include "b2hip_api_callbacks.h"
} // extern "C"
