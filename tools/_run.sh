python -m pytest tests -m gpu -x -q -k "f16x3" 2>&1 | tail -9
run() { python bench.py --steps 100 --warmup 20 --cpu-seconds 0 --graph off "$@" 2>/dev/null | grep "^{" | python -c "
import sys,json
d=json.loads(sys.stdin.read()); x=d.get('f16x3') or {}
print('%-40s wall %.4f ms frac %.3f launch %.4f | x3 %.4f ms launch %.4f dV %.2e same %.4f' % (' '.join(sys.argv[1:]), d['ms_per_step'], d['roofline']['frac'], d['roofline']['launch_ms'], x.get('ms_per_step',0), (x.get('roofline') or {}).get('launch_ms',0), x.get('max_abs_dV_vs_f32_kernels',0), x.get('identical_decisions',0)))" "$@"; }
run --roots 2048
run --humans 30 --roots 1024
