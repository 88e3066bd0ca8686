#!/bin/bash
# A/B of two builds of the library on the GPU box: sibelia_amd/lib/libsibelia_amd.so (A, current sources) against
# sibelia_amd/lib/libsibelia_amd_prev.so (B, a copy kept from the previous build).  Usage: tools/ab_bench.sh OUTDIR [bench args]
out=$1; shift
mkdir -p "$out"
L=sibelia_amd/lib/libsibelia_amd.so
run() { python bench.py --no-cpu-baseline --steps 5 "$@" > "$out/$tag.json" 2> "$out/$tag.err"; python - "$out/$tag.json" "$tag" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], round(d["ms_per_step"], 2), d["config"]["rounds"], d["matches_reference_fixture"], {k: round(v, 2) for k, v in d["phase_ms"].items()})
PY
}
tag=A; run "$@"
if [ -f sibelia_amd/lib/libsibelia_amd_prev.so ]; then cp "$L" "$out/A.so"; cp sibelia_amd/lib/libsibelia_amd_prev.so "$L"; touch "$L"; tag=B; run "$@"; cp "$out/A.so" "$L"; rm -f "$out/A.so"; touch "$L"; fi
