python3 tools/lab_pam.py --reps 3 2>&1 | grep -v amdgpu.ids | tail -4
bash tools/jobs/trace_pam.sh pam11 > /dev/null 2>&1; grep "pairs\|active\|fill" gpurun_out/pam11/kernel_summary.csv; cat gpurun_out/pam11/window.txt | head -30
