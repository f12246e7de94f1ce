cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for cfg in "4096" "4096 batch" "1024"; do
  tag=$(echo $cfg | tr ' ' _)
  rm -rf /tmp/lt; timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/lt -- python3 tools/exp_lml_trace.py $cfg > gpurun_out/lmltrace_$tag.log 2>&1
  python3 tools/exp_lml_trace.py --join /tmp/lt >> gpurun_out/lmltrace_$tag.log 2>&1
done
grep -v amdgpu gpurun_out/lmltrace_4096.log | tail -60
