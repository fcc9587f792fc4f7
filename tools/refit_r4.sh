#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): the round-4 refit measurements — both kernel forms (SGE_BLAS_RAW=0: per-component loads into
# an SoA tile, 1: 16-byte granules into a memory-order tile) on both meshes, and, when the diagnostic variant exists
# (tools/build_variant.sh blasexp4 -DSGE_BLAS_EXPERIMENT=4), the in-kernel phase stamps of both.
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
for raw in 0 1; do
  echo "== SGE_BLAS_RAW=$raw synthetic"; SGE_BLAS_RAW=$raw timeout -k 10 200 python tools/refit_bench.py | grep refit
  echo "== SGE_BLAS_RAW=$raw real";      SGE_BLAS_RAW=$raw timeout -k 10 200 python tools/refit_bench.py --real | grep refit
done 2>&1 | grep -v amdgpu.ids
if [ -f swift-game-engine_amd/libsge_amd_blasexp4.so ]; then
  for raw in 0 1; do
    echo "== phases, SGE_BLAS_RAW=$raw"
    SGE_AMD_LIB=libsge_amd_blasexp4.so SGE_BLAS_RAW=$raw timeout -k 10 200 python tools/refit_phases.py 2>&1 | grep -v amdgpu.ids
  done
fi
