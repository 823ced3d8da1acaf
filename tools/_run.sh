python -m pytest tests -m gpu -x -q 2>&1 | tail -4
run() { python bench.py --steps 200 --warmup 20 --cpu-seconds 0 --graph off "$@" 2>/dev/null | grep "^{" | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('%-60s wall %.4f ms dev %.4f ms frac %.3f launch %.4f' % (' '.join(sys.argv[1:]), d['ms_per_step'], d['step_ms_device']['median'], d['roofline']['frac'], d['roofline']['launch_ms']))" "$@"; }
run --roots 256
RGL_SCENE_EMBED_INSIDE=0 run --roots 256
run --roots 512
RGL_SCENE_EMBED_INSIDE=0 run --roots 512
run --depth 3 --roots 512
RGL_SCENE_EMBED_INSIDE=0 run --depth 3 --roots 512
run --humans 5 --depth 1 --roots 512
RGL_SCENE_EMBED_INSIDE=0 run --humans 5 --depth 1 --roots 512
run --humans 49 --layers 3 --roots 256
RGL_SCENE_EMBED_INSIDE=0 run --humans 49 --layers 3 --roots 256
run --roots 2048 --steps 50
python tools/train_step_time.py
