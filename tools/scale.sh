#!/bin/bash
# Scaling curve of bench.py on ONE node with N = 1, 2, 4, 8 MI355X (BASELINE config 4): one process per GPU over RCCL / xGMI,
# launched exactly as the driver launches it.  Run on a multi-GPU box:   tools/scale.sh [payload] [steps] [warmup]
#   payload: path (theta/phi only, 0.8 MB) | netcorr (237 MB) | full (624 MB: netG + netCorr, BASELINE config 4) — the
#            per-step gradient all-reduce, bucketed (64 MiB) and launched from autograd hooks during backward
# Writes one JSON line per N to gpurun_out/scale_<payload>.jsonl; efficiency is value(N) / (N * value(1)) (weak scaling:
# batch 8 per GPU whatever N).  Expected exchange per step and overlap window: DESIGN.md section 6.
set -u
cd "$(dirname "$0")/.."
PAYLOAD=${1:-full}; STEPS=${2:-20}; WARMUP=${3:-5}
export HSA_ENABLE_IPC_MODE_LEGACY=0          # dmabuf IPC only on this driver (RCCL / cross-process tensor sharing)
OUT=gpurun_out/scale_${PAYLOAD}.jsonl; mkdir -p gpurun_out; : > $OUT
NGPU=$(python -c "import torch; print(torch.cuda.device_count())")
for N in 1 2 4 8; do
  [ "$N" -gt "$NGPU" ] && { echo "only $NGPU GPU(s) visible: stopping at N=$N"; break; }
  PORT=$((29500 + N))
  if [ "$N" = 1 ]; then
    python bench.py --gpus 1 --steps $STEPS --warmup $WARMUP --grad-payload $PAYLOAD --no-cpu-baseline --no-extras | tail -1 >> $OUT
  else
    python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port $PORT \
      bench.py --gpus $N --steps $STEPS --warmup $WARMUP --grad-payload $PAYLOAD --no-cpu-baseline | tail -1 >> $OUT
  fi
done
python - "$OUT" <<'PY'
import json, sys
rows = [json.loads(l) for l in open(sys.argv[1]) if l.startswith("{")]
base = rows[0]["value"] if rows else None
for r in rows:
    ex = r.get("exchange_probe") or {}
    print(f"N={r['n_gpus']}: {r['value']:.0f} images/s, {r['ms_per_step']:.3f} ms/step, efficiency {r['value'] / (r['n_gpus'] * base):.3f}, "
          f"all-reduce bytes/step {r['config']['grad_allreduce_bytes']}, probe {ex}")
PY
