"""dmcf_amd -- MI355X-native implementation of tum-pbs/DMCF's per-step particle hot path.

Fixed-radius neighbour search + ContinuousConv (CConv) + antisymmetric CConv (ASCC) as hand-written
gfx950 HIP kernels behind a C ABI (include/dmcf_hip.h, dmcf_amd/libdmcf_hip.so), exposed through the
reference's own layer / model / pipeline surface:

    dmcf_amd.ops                      <-> open3d.ml.tf ops/layers the reference calls
    dmcf_amd.utils.convolutions       <-> utils/convolutions.py   (ContinuousConv)
    dmcf_amd.utils.tools.losses       <-> utils/tools/losses.py   (window functions, grid_pos)
    dmcf_amd.models                   <-> models/                 (PBFNet, HRNet, SymNet, CConv)
    dmcf_amd.pipelines                <-> pipelines/              (Simulator.run_inference / run_rollout)
"""
__version__ = "0.1.0"
