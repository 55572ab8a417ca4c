cd $GRAFT_REPO_ROOT
pk() { python3 -c "
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1], 'ms/step %.3f'%d['ms_per_step'], {k:round(v['ms'],3) for k,v in d['kernels'].items() if k.startswith('mlp')})" $1; }
for rep in 1 2; do
touch project-nerf_amd/csrc/mlp_wgrad.hip
timeout 900 python3 project-nerf_amd/build.py -q >/dev/null
timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-render > gpurun_out/b_plain$rep.json 2>/dev/null; pk gpurun_out/b_plain$rep.json
touch project-nerf_amd/csrc/mlp_wgrad.hip
NERF_EXTRA_CXXFLAGS=-DNERF_WGRAD_NT_ON timeout 900 python3 project-nerf_amd/build.py -q >/dev/null
timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-render > gpurun_out/b_nt$rep.json 2>/dev/null; pk gpurun_out/b_nt$rep.json
done
