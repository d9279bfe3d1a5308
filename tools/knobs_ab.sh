#!/bin/bash
# Same-box A/B of launch knobs (one gpurun call: boxes differ by a few %).  Prints value / roofline.frac / trip ms / host issue ms per frame.
# Usage: tools/knobs_ab.sh "label|ENV=1 ENV2=2|--extra args" ...   (default: the production setting twice)
run() {
  label=$1; shift
  out=$(env "$@" 2>/dev/null | tail -1)
  python - "$label" <<P
import json,sys
d=json.loads('''$out''')
print(sys.argv[1], d["value"], "frac", d["roofline"]["frac"], "trips_ms", d["roofline"]["ms_per_frame_all_trips"], "issue_ms", d["config"].get("host_issue_ms_per_frame"), "ms/frame", d["ms_per_step"])
P
}
B="python bench.py --no-configs --no-modes --no-cpu-baseline --no-grid-stage --long-run-frames 0"
[ $# -eq 0 ] && set -- "default||" "default again||"
for spec in "$@"; do
  IFS='|' read -r label envs extra <<< "$spec"
  run "$label" $envs $B $extra
done
