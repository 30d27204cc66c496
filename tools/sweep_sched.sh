# Sweep of the scheduler thresholds on the GPU box (kernel ms by HIP events, 64 passes at 1080p):
#   bash tools/sweep_sched.sh            (RT_REFILL_BELOW x scene, SAH + culled; then the reference structure)
for sc in suzanne dragon suzanne5; do
for r in 8 12 16 20 24; do
p=64; [ $sc = suzanne5 ] && p=16
echo -n "sah $sc refill $r: "; RT_SCENE=$sc RT_REFILL_BELOW=$r RT_SAH=1 RT_CULL=1 timeout -k 10 200 python tools/prof_run.py $p | tail -n 1 | sed 's/.*kernel_ms=//' || exit 1
done; done
for r in 16 20 24 28 32; do
echo -n "ref suzanne refill $r: "; RT_REFILL_BELOW=$r RT_CULL=1 timeout -k 10 100 python tools/prof_run.py 64 | tail -n 1 | sed 's/.*kernel_ms=//' || exit 1
done
for r in 24 28 32 36; do
echo -n "ref strict suzanne refill $r: "; RT_REFILL_BELOW=$r timeout -k 10 100 python tools/prof_run.py 16 | tail -n 1 | sed 's/.*kernel_ms=//' || exit 1
done
