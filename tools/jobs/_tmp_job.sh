LAB_CONFIGS="1,1,-1,1" python3 tools/lab_pass.py enspara_amd/_variants/stamps.so --centers 3000 2>&1 | grep -v amdgpu.ids | grep chain | tail -3
LAB_CONFIGS="1,1,-1,1;1,1,-1,0;1,0,1,0" python3 tools/lab_pass.py --centers 5000 2>&1 | grep -v amdgpu.ids | tail -4
python -m pytest tests -q -m gpu -x 2>&1 | tail -2
