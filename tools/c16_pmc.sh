# effective clock / MFMA-pipe utilisation of the fp16 tower conv timing variants (see tools/c16_x.sh)
export TMPDIR=/tmp
for d in ${DS:-0 32 1 15}; do
  D=gpurun_out/c16pmc_$d; rm -rf $D
  AGZ_C16_DEBUG=$d rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES --kernel-trace --output-format csv -d $D -- python tools/nn_micro.py --batches 8192 --precision f16 --iters 2 > $D.log 2>&1
  echo "DEBUG=$d"; python tools/pmc_mfma.py $D | grep -v "^kernel" | cut -d, -f1,2,3,4,5,9,10
  find $D -name '*.csv' -size +2M -delete
done
