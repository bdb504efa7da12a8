#!/bin/bash
# On the GPU box (scratch copy of the repo): every variants/libdpf_<name>.so in turn over dpf_nets_amd/libdpf_hip.so, the reproducer of
# profiles/r06_emd_bisect.txt through launches K0..K1 (default 6..9: level 6's passes), R runs each.   run.sh [R [K0 [K1 [names...]]]]
R=${1:-16}; K0=${2:-6}; K1=${3:-9}; shift 3 2>/dev/null
cd "$(dirname "$0")/../.." || exit 1
cp dpf_nets_amd/libdpf_hip.so /tmp/libdpf_shipped.so
names=("$@"); if [ ${#names[@]} -eq 0 ]; then for f in variants/libdpf_*.so; do n=${f#variants/libdpf_}; names+=("${n%.so}"); done; fi
for n in "${names[@]}"; do
  echo "== $n"
  cp "variants/libdpf_$n.so" dpf_nets_amd/libdpf_hip.so
  timeout 600 python3 tests/diag/emd_bisect.py 2 128 64 692 "$R" "$K0" "$K1" 2>&1 | tail -n $((K1 - K0 + 2))
done
cp /tmp/libdpf_shipped.so dpf_nets_amd/libdpf_hip.so
