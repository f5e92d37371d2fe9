#!/bin/bash
# build in-tree (CPU, cross-compiles gfx950), then run a command on the MI355X box
set -e
cd "$(dirname "$0")/.."
python -m raytracer_amd.build >/dev/null
make -s -C oracle oracle ref >/dev/null 2>&1
T=${GPU_TIMEOUT:-600}
exec /usr/local/graft/bin/gpurun --timeout $T -- "$@"
