cd /root/repo
for v in ${ABL_LIST:-0 671 1695 2463}; do
  lib=variants/libcaelo_abl$v.so; [ $v = 0 ] && lib=cae-lo_amd/caelo/libcaelo.so
  for n in ${N_LIST:-768 3072 12288 24576 49152}; do
    echo "== ABL $v n $n: $(S1X_N=$n CAELO_LIB=$PWD/$lib python tools/stage1_density_sweep.py 0.0 2>&1 | tail -1)"
  done
done
