#!/bin/bash
# GPU box: the five BASELINE configurations in full through the command-line renderer, one GPU; timings on stderr.
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
R="timeout -k 10 250 python -m raytracer_amd.render"
$R --scene 1 --width 1200 --spp 100 --out gpurun_out/c1.png 2>&1 | tail -1 || exit 1
$R --scene 7 --width 600 --spp 1000 --out gpurun_out/c2.png 2>&1 | tail -1 || exit 1
$R --scene 8 --width 600 --spp 1000 --out gpurun_out/c3.png 2>&1 | tail -1 || exit 1
$R --scene 9 --width 800 --spp 10000 --out gpurun_out/c4.png 2>&1 | tail -1 || exit 1
$R --scene 12 --triangles 10000000 --width 3840 --height 2160 --spp 256 --bvh lbvh --out gpurun_out/c5.png 2>&1 | tail -2 || exit 1
