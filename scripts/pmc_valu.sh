cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/final
for rend in megakernel wavefront; do
timeout -k 5 150 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/final/pmc_${rend}_valu -- python3 bench.py --steps 1 --warmup 0 --spp 4 --no-cpu-baseline --no-pipelined --renderer $rend > gpurun_out/final/pmc_${rend}_valu.log 2>&1; echo "valu $rend rc=$?"
timeout -k 5 150 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_SALU --output-format csv -d gpurun_out/final/pmc_${rend}_wave -- python3 bench.py --steps 1 --warmup 0 --spp 4 --no-cpu-baseline --no-pipelined --renderer $rend > gpurun_out/final/pmc_${rend}_wave.log 2>&1; echo "wave $rend rc=$?"
done
python3 scripts/pmc_summarize.py gpurun_out/final/pmc_megakernel_valu gpurun_out/final/pmc_megakernel_wave gpurun_out/final/pmc_wavefront_valu gpurun_out/final/pmc_wavefront_wave | grep -E "megakernel|extend|shade"
