export EG3D_LIB=$PWD/edgegraph3d_amd/variants/libeg3d_timing.so
python tools/section_timing.py 3 > gpurun_out/st_c3.txt 2>&1
python tools/section_timing_c4.py 1500 > gpurun_out/st_c4.txt 2>&1
python tools/section_timing.py 2 > gpurun_out/st_c2.txt 2>&1
cat gpurun_out/st_c3.txt gpurun_out/st_c4.txt gpurun_out/st_c2.txt
