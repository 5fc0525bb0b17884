for p in 1 2 3 0; do
echo "PROBE $p"
PFSLAM_PROBE=$p PFSLAM_CELLS_MODE=1 bash tools/timeline.sh 2>&1 | grep -E "frames 20|k_cells_update|k_cells_mark"
done
