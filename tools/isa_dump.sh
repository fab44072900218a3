#!/bin/bash
# Dump the gfx950 ISA of csrc/scope_kernels.hip to $1 (default /tmp/scope_kernels.s).
set -e
root="$(cd "$(dirname "$0")/.." && pwd)"
out="${1:-/tmp/scope_kernels.s}"
cd "$root/obs-color-monitor_amd/csrc"
hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I. -S --cuda-device-only -o "$out" scope_kernels.hip 2>&1 | grep -v "warning: argument unused" || true
echo "$out"
