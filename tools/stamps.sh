#!/bin/bash
# Diagnostic (never the product): build the library with -DHHE_STAMPS (ks_row_kernel records s_memtime at its phase boundaries),
# run one bench step through it and print the per-phase timeline of the sampled workgroups.  Usage: tools/stamps.sh [extra hipcc flags]
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
csrc=$root/privacy-preserving-ml-through-hhe_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DHHE_STAMPS "$@" -o $root/tools/libhhe_stamps.so $csrc/hhe_kernels.hip $csrc/hhe_api.cpp $csrc/hhe_context.cpp \
    $csrc/hhe_pasta_public.cpp $csrc/hhe_client.cpp $csrc/hhe_seal_wire.cpp
