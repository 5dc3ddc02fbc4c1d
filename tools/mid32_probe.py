import sys, os
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import raytracer_amd as R
from raytracer_amd import runtime
src = open('/root/repo/tests/test_gpu_parity.py').read()
i = src.index('def _crowded_scene_text'); j = src.index('@pytest.mark.parametrize("grid"')
ns = {}; exec(src[i:j], ns)
text = ns['_crowded_scene_text'](22, 8, seed=3).replace("set 0.3 0.4 0.6 2 3", "set 0.3 0.4 0.6 8 3")
open('gpurun_out/mid32.txt', 'w').write(text)
sc = R.parse_scene_file('gpurun_out/mid32.txt')
r = runtime.Renderer(0); r.set_scene(sc)
for rep in range(4):
    _, u8, st = r.render_rows(1920, 1080, seed=rep, want_f32=False)
print(len(sc.surfaces), 'surfaces: trace %.3f shade %.3f (main %.3f) ms' % (st['ms_trace'], st['ms_shade'], st['ms_shade_main']), st['wave_light'])
