for rep in 1 2 3; do for f in "" "-DPF_NO_BIG_THETA"; do PFSLAM_EXTRA_FLAGS="$f" python tools/experiments/r04/score_pass_ab.py 2>/dev/null | tail -1; done; done
