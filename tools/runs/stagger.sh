for v in base stg3 stg6 base stg3 stg6; do
  if [ $v = base ]; then unset MADELEINE_LIB; else export MADELEINE_LIB=$GRAFT_REPO_ROOT/tools/ab/$v.so; fi
  python tools/exp_nt_shapes.py 2>&1 | grep "^M" | head -3 | sed "s/^/$v /"
done
unset MADELEINE_LIB
bash tools/runs/r06_ab.sh stg3 stg6
