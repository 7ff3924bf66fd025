python3 tools/pam_rt2.py > gpurun_out/b3.json 2> gpurun_out/b3.err; grep -v amdgpu gpurun_out/b3.err | head -45
