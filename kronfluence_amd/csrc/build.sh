#!/usr/bin/env bash
# Builds libkronfluence_hip.so for gfx950 in-tree (the .so travels to the GPU box with the snapshot).
set -euo pipefail
here="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
out="${here}/../libkronfluence_hip.so"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
flags=(--offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wall -Wno-unused-function)
mkdir -p "${here}/obj"
pids=()
for src in kf_kernels kf_eigh kf_score_v2; do
  stale=0
  [[ -f "${here}/obj/${src}.o" ]] || stale=1
  for dep in "${here}/${src}.hip" "${here}"/*.h "${here}/../../include/kronfluence_hip.h"; do
    [[ "${dep}" -nt "${here}/obj/${src}.o" ]] && stale=1
  done
  if [[ ${stale} -eq 1 ]]; then
    "${HIPCC}" "${flags[@]}" -c "${here}/${src}.hip" -o "${here}/obj/${src}.o" &
    pids+=($!)
  fi
done
for p in "${pids[@]:-}"; do [[ -n "${p}" ]] && wait "${p}"; done
"${HIPCC}" --offload-arch=gfx950 -shared -fPIC "${here}/obj/kf_kernels.o" "${here}/obj/kf_eigh.o" "${here}/obj/kf_score_v2.o" -o "${out}"
echo "built ${out}"
