#!/bin/bash
# Dump the gfx950 ISA of csrc/scope_trio.hip to /tmp/trio.s and the kernel $1 (default <7, true, 32>) alone to /tmp/trio7.s.
# Extra arguments go to hipcc (e.g. -DTRIO_RLE=0).
root="$(cd "$(dirname "$0")/.." && pwd)"
k="${1:-_Z17scope_trio_kernelILi7ELb1ELi32EE}"; shift
hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I"$root/obs-color-monitor_amd/csrc" "$@" -S --cuda-device-only -o /tmp/trio.s \
  "$root/obs-color-monitor_amd/csrc/scope_trio.hip" 2>&1 | grep -E "error" | head
L=$(grep -n "^$k" /tmp/trio.s | head -1 | cut -d: -f1)
awk -v l="$L" 'NR>=l' /tmp/trio.s | awk '{print} /^\.Lfunc_end/{exit}' > /tmp/trio7.s
wc -l /tmp/trio7.s
awk -v l="$L" 'NR>=l' /tmp/trio.s | grep -m3 -E "NumVgprs|ScratchSize|Occupancy"
