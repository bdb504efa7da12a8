for i in 1 2 3; do for so in libdpf_c1.so libdpf_c2.so; do python tools/lib_ab.py $so --no-cpu-baseline --no-extra --steps 400 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$so', round(d['ms_per_step']*1e3,2), {k:round(v,2) for k,v in d['roofline']['kernels_us'].items()})"; done; done
