#!/bin/bash
# Socket power and clocks while the predict kernel runs back to back (GPU box): bash tools/power_probe.sh [mode]
# Evidence for DESIGN section 8's "the chip holds its clock down": rocm-smi samples before, during and after a 20 s run.
mode=${1:-f16x3}
out=gpurun_out/power_${mode}.txt
mkdir -p gpurun_out
{
echo "== idle"; rocm-smi --showpower --showclocks --showmaxpower --showtemp 2>&1 | grep -E "Power|sclk|mclk|Temperature \(Sensor (junction|hotspot)|Max" ;
} > $out
python - "$mode" >> $out 2>&1 <<'PY' &
import sys, time, os
sys.path.insert(0, os.getcwd())
import torch, bench
import seq2squiggle_amd as S
mode = sys.argv[1]
sd, cfg = S.load_checkpoint(os.path.join("tests", "golden", "synthetic_k9.ckpt"))
eng = S.Engine(sd, cfg, device=0, mode=mode)
reads = bench.make_reads(1000, 1234)
bases, nv, first = S.encode_reads(reads, cfg["seq_kmer"])
b, n = torch.from_numpy(bases).to(eng.device), torch.from_numpy(nv).to(eng.device)
sig = torch.empty(b.shape[0], 250, dtype=torch.float32, device=eng.device); dur = torch.empty(b.shape[0], 16, dtype=torch.int32, device=eng.device)
p = S.PredictParams(seed=42)
t0 = time.perf_counter(); k = 0
while time.perf_counter() - t0 < 22:
    eng.predict_chunks(b, n, p, out_signal=sig, out_dur=dur); k += 1
    if k % 8 == 0: torch.cuda.synchronize()
torch.cuda.synchronize()
el = time.perf_counter() - t0
print(f"== {mode}: {k * b.shape[0] / el:.4e} chunks/s over {el:.1f} s")
PY
pid=$!
sleep 8
for i in 1 2 3 4 5; do
  echo "== under load, sample $i" >> $out
  rocm-smi --showpower --showclocks --showtemp 2>&1 | grep -E "Power|sclk|mclk|Temperature \(Sensor (junction|hotspot)" >> $out
  sleep 2
done
wait $pid
cat $out
