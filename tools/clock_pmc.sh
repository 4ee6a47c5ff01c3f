# Measures the shader clock under load and the matrix-pipe utilisation of the two dominant conv kernels:
#   clock = GRBM_GUI_ACTIVE (summed over the 8 XCDs) / 8 / kernel duration;  MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (SQ_BUSY_CU_CYCLES x 4 SIMDs)
# Round 1: 2.05 GHz inside winograd_fused_kernel, 2.15-2.24 GHz inside conv_mfma_dma_kernel<128,128> (nominal 2.4 GHz).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d $R/gpurun_out/pmc_clk_wf -- python3 $R/tools/wf_one.py 128 256 256 > $R/gpurun_out/pmc_clk_wf.log 2>&1 || echo FAILED1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d $R/gpurun_out/pmc_clk_dc -- python3 $R/tools/one_conv.py --n 16 --hw 64 --cin 512 --c2 512 --cout 512 --k 1 > $R/gpurun_out/pmc_clk_dc.log 2>&1 || echo FAILED2
cd $R && python3 tools/mfma_util.py gpurun_out/pmc_clk_wf gpurun_out/pmc_clk_dc gpurun_out/mfma_util.json
