#!/bin/bash
# Same-box A/B: check out <ref> (default HEAD) into .ab_prev/ (git-ignored, travels with the gpurun snapshot) and build it,
# so that one gpurun call can time `python .ab_prev/bench.py ...` and `python bench.py ...` back to back on ONE box
# (boxes of the pool differ by ~3 % on the step).   usage: scripts/ab_prev.sh [ref]
ref=${1:-HEAD}
root=$(git rev-parse --show-toplevel)
cd $root
if [ -d .ab_prev ]; then git worktree remove --force .ab_prev; fi
git worktree add --detach .ab_prev $ref > /dev/null
make -C .ab_prev/dusty_gan_amd/csrc -j8 > /dev/null && echo "built .ab_prev @ $(git -C .ab_prev rev-parse --short HEAD)"
