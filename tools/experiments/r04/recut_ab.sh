OVENV=PFSLAM_CELLS_MODE=1 OVCUT=12 bash tools/experiments/r04/overlaps.sh | tr '\n' ' '; echo
bash tools/experiments/r04/repeat.sh 5 X=1; bash tools/experiments/r04/repeat.sh 3 PFSLAM_CELLS_RECUT_EVERY=0; bash tools/experiments/r04/repeat.sh 3 PFSLAM_CELLS_RECUT_EVERY=4
