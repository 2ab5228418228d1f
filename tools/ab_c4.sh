for lib in "" edgegraph3d_amd/variants/libeg3d_m3k3.so edgegraph3d_amd/variants/libeg3d_m3k2.so; do
  [ -n "$lib" ] && export EG3D_LIB=$PWD/$lib || unset EG3D_LIB
  python bench.py --workload c4 --steps 2 --warmup 1 --inflight 1 --no-cpu-baseline --no-extras 2>/dev/null | tail -n 1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$lib c4', round(d['value']), round(d['ms_per_step'],1), d['stage_ms'])"
done
