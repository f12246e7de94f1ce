cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 | tee gpurun_out/t1_pytest.log
timeout -k 10 300 python tools/exp_balanced.py 1024 4096 8192 10112 12288 2>&1 | grep -v amdgpu.ids | cut -c1-250 | tee gpurun_out/t1_bal.log
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for cfg in "4096" "4096 batch"; do
  tag=$(echo $cfg | tr ' ' _)
  rm -rf /tmp/lt; timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/lt -- python3 tools/exp_lml_trace.py $cfg > gpurun_out/lmltrace_$tag.log 2>&1
  python3 tools/exp_lml_trace.py --join /tmp/lt >> gpurun_out/lmltrace_$tag.log 2>&1
  grep -v "amdgpu\|rocprofv3\|tool.cpp\|output_stream" gpurun_out/lmltrace_$tag.log | tail -22
done
