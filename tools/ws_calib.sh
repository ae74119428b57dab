#!/bin/bash
# Round-5 calibration of the ag_gemm_ws planner: the masked forward at small batches with each Linear pinned to a route (AG_WS_FORCE =
# "N:K:route:splits;..."; routes: 0 ag_gemm, 1 ag_gemm_resid_split, 2 128-tile units, 3 128-tile units x splits + row kernel).
# usage: bash tools/ws_calib.sh <out.jsonl> <point> [<point> ...]     (points as tools/fwd_points.py)
OUT=$1; shift
run() {  # tag, force
  FP_TAG="$1" AG_WS_FORCE="$2" FP_STEPS=20 python tools/fwd_points.py "${@:3}" 2>/dev/null >> $OUT
}
H=${WS_H:-768}; I=${WS_I:-3072}; Q=$((3*H))
AG_WS_ROUTE=0 FP_TAG=off FP_STEPS=20 python tools/fwd_points.py "$@" 2>/dev/null >> $OUT
run auto "" "$@"
run all_ex "$Q:$H:2:0;$H:$H:2:0;$I:$H:2:0;$H:$I:2:0" "$@"
run o_ex "$Q:$H:0:0;$H:$H:2:0;$I:$H:2:0;$H:$I:1:0" "$@"
run o_slab2 "$Q:$H:0:0;$H:$H:3:2;$I:$H:0:0;$H:$I:1:0" "$@"
run fc2_slab2 "$Q:$H:0:0;$H:$H:0:0;$I:$H:0:0;$H:$I:3:2" "$@"
run fc2_slab4 "$Q:$H:0:0;$H:$H:0:0;$I:$H:0:0;$H:$I:3:4" "$@"
run fc2_ex "$Q:$H:2:0;$H:$H:0:0;$I:$H:0:0;$H:$I:2:0" "$@"
run fc1_ex "$Q:$H:0:0;$H:$H:2:0;$I:$H:2:0;$H:$I:0:0" "$@"
run qkv_ex "$Q:$H:2:0;$H:$H:0:0;$I:$H:0:0;$H:$I:0:0" "$@"
