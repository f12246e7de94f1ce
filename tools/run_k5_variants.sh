for v in base ord1 ord2 nt sc1 base; do
  if [ $v = base ]; then unset GPK_LIBRARY; else export GPK_LIBRARY=$GRAFT_REPO_ROOT/unmanned_aerial_vehicles_amd/build/libgpk_$v.so; fi
  echo "== $v"
  SWEEP=0 FORMS=direct:24,direct:24 REPS=8 timeout -k 10 200 python tools/exp_k5_direct.py 2>&1 | grep direct
done
