#!/bin/bash
run() { echo "== $*"; env "$@" DBG_SYNC=stream timeout 200 python scripts/debug_seg_sync2.py 2>&1 | grep -v amdgpu.ids | grep " [4567] graph" | cut -c1-100; }
run DUSTY_GAN_FUSE_PROJ=0
run DUSTY_GAN_FUSE_PROJ=0 A=2
run DG_PROJ_ADAM_MFMA=0
run DBG_B=32
run DBG_B=16
run DBG_B=8 DBG_FP32=1
