#!/bin/bash
# builds the library and its profiling twin; fails if either does not compile
set -e
cd /root/repo/mcmc_amd/csrc
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -I../../include -o ../lib/libmcmc_amd.so mhd_api.hip 2>&1 | grep -E "error|Illegal" -A5 && exit 1
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DMHD_PROFILE -I../../include -o ../lib/libmcmc_amd_prof.so mhd_api.hip 2>&1 | grep -E "error|Illegal" -A5 && exit 1
echo built
