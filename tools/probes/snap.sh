#!/bin/bash
# usage: tools/probes/snap.sh NAME   -- snapshot of the kernel sources as they are now, for ab.sh:
#        tools/probes/ab.sh NAME=variants/NAME/lzs_kernels.hip
cd "$(dirname "$0")"
rm -rf variants/$1 && mkdir -p variants/$1
cp ../../lzs_compression_amd/csrc/lzs_kernels.hip ../../lzs_compression_amd/csrc/lzs_hip_shim.h variants/$1/
cp -r ../../lzs_compression_amd/csrc/kernels variants/$1/
