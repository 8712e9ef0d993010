set -u
mkdir -p gpurun_out
timeout 1150 python -m pytest tests -m gpu -x -q > gpurun_out/r2s2_gputests.log 2>&1; echo "pytest rc=$?"
grep -v amdgpu.ids gpurun_out/r2s2_gputests.log | tail -5
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -2
