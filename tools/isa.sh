#!/bin/bash
# usage: tools/isa.sh OUT.s [extra hipcc flags]  -- the kernels' ISA with line tables (for tools/isa_blocks.py, isa_lines.py)
out=$1; shift
cd "$(dirname "$0")/../lzs_compression_amd/csrc" && /opt/rocm/bin/hipcc -Os -fno-unroll-loops -fPIC --offload-arch=gfx950 -gline-tables-only -Wall -Wno-unused-function -I. -S --cuda-device-only "$@" lzs_kernels.hip -o "$out" 2>&1 | grep -v "hip-link"
