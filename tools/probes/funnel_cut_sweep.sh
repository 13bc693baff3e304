for c in 5 7; do export CMCD_LIB_PATH=$GRAFT_REPO_ROOT/variants/libcmcd_hip_cut$c.so; echo "cut $c"; python tools/probes/funnel_ab.py 2>/dev/null | grep "variant 4"; done
unset CMCD_LIB_PATH; echo "cut 6 (product)"; python tools/probes/funnel_ab.py 2>/dev/null | grep "variant 4"
