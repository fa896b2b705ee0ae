#!/bin/bash
# The randomised parity sweep as a tool over every seed used in the round (GPU box): one JSON report per line into $1
# (default gpurun_out/fuzz_parity.log).  89 runs x 40 scenes (~13 s each) + 40 runs x 20 large-rig scenes (~17 s each).  The suite asserts the same comparison on ten of the seeds
# (tests/test_gpu_fuzz_parity.py); the criteria are tests/parity.py's.
OUT=${1:-gpurun_out/fuzz_parity.log}
mkdir -p "$(dirname "$OUT")"
: > "$OUT"
# PAR (default 1): runs side by side (each is one host thread of oracle + numpy work around small launches; round 6: PAR=10, ~5 minutes for the sweep)
PAR=${PAR:-1}
run() {   # seed trials [ENV=value ...]
  local s=$1 n=$2; shift 2
  env "${@:-_X=0}" timeout 600 python tools/fuzz_parity.py $s $n 2>> "$OUT.err.$s" | tail -1 > "$OUT.part.$s"
}
export -f run; export OUT
SMALL="4 15 37 39 0 1 2 3 $(seq -s ' ' 101 112) $(seq -s ' ' 201 206) $(seq -s ' ' 301 340)"
{ for s in $SMALL; do echo "$s 40"; done
  for s in 5 $(seq 341 350); do echo "$s 40 FUZZ_POSE=h36m"; done
  for s in 6 $(seq 351 360); do echo "$s 40 FUZZ_NORM_HEIGHT=1"; done
  echo "7 40 FUZZ_POSE=h36m FUZZ_NORM_HEIGHT=1"
  # large rigs (32 / 48 cameras, 12 - 24 persons: rows pair kernel with column chunks, per-camera sweep, per-joint triangulation, per-step stage 2)
  for s in $(seq 1001 1040); do echo "$s 20 FUZZ_LARGE=1"; done; } > "$OUT.jobs"
xargs -P $PAR -L 1 bash -c 'run "$@"' _ < "$OUT.jobs"
while read s rest; do cat "$OUT.part.$s" >> "$OUT"; cat "$OUT.err.$s" >> "$OUT.err" 2>/dev/null; rm -f "$OUT.part.$s" "$OUT.err.$s"; done < "$OUT.jobs"
rm -f "$OUT.jobs"
python3 - "$OUT" <<'PY'
import json, sys
rows = [json.loads(l) for l in open(sys.argv[1]) if l.startswith("{")]
bad = [r["seed"] for r in rows if r["joints_over_1mm"] or r["scores_over_rtol"] or r["cov_over_tol"] or r["ill_not_listed"] or r["stage2_numeric_failures"] or r["stage2_index_divergences"]]
tot = lambda k: sum(r[k] for r in rows)
print("runs %d, joints %d (excluded %d; second pass %d; input-derived kappa >= 2e3: %d, of them not listed %d), over 1 mm %d (worst %.4f mm), scores %d, covariances %d (over the flat 1e-3: %d, worst %.2e), sequence-steps %d (excluded %d), stage-2 numeric %d, index %d; runs with a miss: %s"
      % (len(rows), tot("joints"), tot("joints_excluded"), tot("joints_second_pass"), tot("ill_joints"), tot("ill_not_listed"), tot("joints_over_1mm"), max(r["pos_max_mm"] for r in rows), tot("scores_over_rtol"), tot("cov_over_tol"), tot("cov_over_flat_1e-3"),
         max(r["cov_rel_max"] for r in rows), tot("seq_steps"), tot("seq_steps_excluded"), tot("stage2_numeric_failures"), tot("stage2_index_divergences"), bad))
PY
