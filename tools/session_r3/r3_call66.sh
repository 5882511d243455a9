for v in 0 1 2 3 4 5 6; do DISYOLO_QUAD_TILE=$v python tools/bench_quad.py 2>&1 | grep -v amdgpu.ids; done
