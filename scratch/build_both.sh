#!/bin/bash
# builds the library, its test twin and its profiling twins with the package's own recipe
set -e
cd "$(dirname "$0")/.." && python -m mcmc_amd.build --force --profile
echo built
