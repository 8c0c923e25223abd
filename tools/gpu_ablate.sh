# usage: bash tools/gpu_ablate.sh  — same-box ablation of the hidden-layer forward (RB_FWD2_ABLATE bits: 1 no weight
# refill, 2 no activation refill, 4 no MFMA); prints the HIP-event average of the fc_h_fwd launch for each variant.
mkdir -p gpurun_out
rm -f gpurun_out/ablate.log
for A in ${ABLS:-0 1 2 3 4 7 0}; do
  RB_FWD2_ABLATE=$A timeout 300 python bench.py --steps 600 --warmup 100 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); o=[r for r in [d['roofline']]+d.get('roofline_others',[]) if r['kernel']=='fc_h_fwd'][0]; print('ablate=$A fc_h_fwd', round(o['avg_us'],2), 'us   step', round(d['ms_per_step']*1000,1),'us')" >> gpurun_out/ablate.log
done
cat gpurun_out/ablate.log
