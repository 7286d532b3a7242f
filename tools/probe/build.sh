#!/bin/bash
# Builds the micro-probes of this directory for gfx950 next to their sources (binaries are git-ignored).
set -e
cd "$(dirname "$0")"
for f in *.hip; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o "${f%.hip}" "$f"
done
