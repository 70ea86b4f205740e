# ms per step of the N = 100 000 headline step under every noise setting x feature scale x data law: looks for cliffs (a guess that fails
# everywhere, a list that overflows) rather than for speed.  [NODES=20000] bash tools/regime_matrix.sh  (GPU box)
for data in randn clustered; do
for nz in ranked rsym none hash sym; do
  for fs in 0.25 0.5 1 2 4; do
    out=$(timeout 300 python3 bench.py ${NODES:+--nodes $NODES} --noise $nz --feat-scale $fs --data $data --steps 5 --warmup 2 --repeats 1 --cpu-rows -1 --no-variants --no-configs 2>/dev/null | tail -1)
    ms=$(echo "$out" | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.3f' % d['ms_per_step'])" 2>/dev/null || echo "FAILED")
    echo "data $data noise $nz feat-scale $fs: $ms ms/step"
  done
done
done
