#!/bin/bash
# The pipeline at BASELINE configs[1]'s shape (Qwen2.5-0.5B, bs 512 x seq 2048) on ONE MI355X in three topologies:
# 1 learner + 1 engine, 2 + 2 with HIP-IPC weights, 2 + 2 with the gloo weight-update group (the configs[2] topology where RCCL cannot run).
# usage: gpurun --timeout 1800 -- 'bash scripts/gpu_pipeline.sh r06g [steps]'
set -u
TAG=${1:-pipeline}
STEPS=${2:-3}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
for cfg in "1 1 ipc" "2 2 ipc" "2 2 gloo"; do
  set -- $cfg
  name=pipeline_0p5b_${1}x${2}_$3
  ( time timeout 900 python scripts/pipeline_cfg1.py --steps $STEPS --learners $1 --engines $2 --weights $3 --stacks-after 600 --out $OUT/$name.json ) > $OUT/$name.log 2> $OUT/$name.err
  echo "$name exit $?"
  python - "$OUT/$name.json" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read())
except Exception as e:
    print(" no result", e); sys.exit(0)
if "error" in d:
    print(" ERROR", json.dumps(d["error"])[:1500]); sys.exit(0)
s = d["summary"]
print(" samples/s", round(s["samples_per_s"], 2), "s/step", round(s["s_per_step"], 2), "busy", {k: round(v, 3) for k, v in s["busy_frac"].items()},
      "wsync under load ms", s["weight_sync_under_load_ms"], "topology", s.get("topology"), "engines equal trainer", s["engine_weights_equal_trainer_at_last_version"])
PY
done

if [ "${3:-}" == "7b" ]; then
  # BASELINE configs[2]'s topology AT THE 7B SHAPE on one GPU (reduced batch: a 4096-sample step would take half an hour): two 7.6 B-parameter
  # learners (bf16 body, fp32 head, AdamW, gradient checkpointing) + two engines holding their own 16.3 GB copies, HIP-IPC hand-off to both
  name=pipeline_7b_2x2_ipc_bs16_seq8192
  ( time timeout 1500 python scripts/pipeline_cfg1.py --model 7b --global-batch 16 --seq-length 8192 --steps 3 --gradient-checkpointing --learners 2 --engines 2 \
      --weights ipc --stacks-after 1200 --timeout 1400 --out $OUT/$name.json ) > $OUT/$name.log 2> $OUT/$name.err
  echo "$name exit $?"
  python - "$OUT/$name.json" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read())
except Exception as e:
    print(" no result", e); sys.exit(0)
if "error" in d:
    print(" ERROR", json.dumps(d["error"])[:3000]); sys.exit(0)
s = d["summary"]
print(" samples/s", round(s["samples_per_s"], 3), "s/step", round(s["s_per_step"], 2), "busy", {k: round(v, 3) for k, v in s["busy_frac"].items()},
      "wsync under load ms", s["weight_sync_under_load_ms"], "peak GB", s["learner_peak_memory_GB"], "engines equal trainer", s["engine_weights_equal_trainer_at_last_version"])
PY
fi
if [ "${3:-}" == "topologies" ]; then
  # BASELINE configs[3] (4 learners + 4 engines) and configs[4] (4 learners + 2 x TP2 engines, KL on) as TOPOLOGIES on one GPU, two-layer model:
  # ten / eight processes, real kernels, gloo gradients; IPC hand-off to four engines / per-TP-rank gloo groups with sharded updates
  for cfg in "cfg3_4x4_ipc --learners 4 --engines 4 --weights ipc" "cfg4_4x2xTP2_gloo_kl --learners 4 --engines 2 --engine-tp 2 --weights gloo --kl-coef 0.001"; do
    set -- $cfg
    name=pipeline_tiny_$1; shift
    ( time timeout 900 python scripts/pipeline_cfg1.py --model tiny --global-batch 32 --seq-length 128 --attempts 4 --steps 3 "$@" --stacks-after 600 --out $OUT/$name.json ) > $OUT/$name.log 2> $OUT/$name.err
    echo "$name exit $?"
    python - "$OUT/$name.json" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read())
except Exception as e:
    print(" no result", e); sys.exit(0)
if "error" in d:
    print(" ERROR", json.dumps(d["error"])[:3000]); sys.exit(0)
s = d["summary"]
print(" steps", s["optimizer_steps"], "topology", {k: s["topology"][k] for k in ("learners", "engines", "engine_tp", "grad_backend", "weight_transport", "micro_batches_per_learner", "updates_per_engine")},
      "wsync ms", round(s["weight_sync_under_load_ms"]["median"], 1), "engines equal trainer", s["engine_weights_equal_trainer_at_last_version"])
PY
  done
fi
