# the skewed, tie-heavy C5-shaped index of tools/sweep_txh.py (k-means of a few iterations over clustered data:
# leaves of 0 .. 88 k points, thousands of points per approximate distance): the device path must not
# overflow its candidate buffers
for v in 1 2 3 4 5 6; do
timeout -k 10 300 python3 tools/sweep_txh.py --num-points 12500000 --dim 96 --S 24 --leaves 1250 --Ps 10 --ms 1000,8192 --steps 10 2>&1 | grep "P=\|Error" | cut -c1-120
done
