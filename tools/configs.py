"""Model sections of the reference's shipped configurations, restated as Python dicts so that tests and
bench.py run where /root/reference does not exist (the GPU box).  tests/test_models_cpu.py checks them
against the YAML files whenever the reference tree is present.  Source: configs/<name>.yml, section ``model``
(keys that only matter for training -- loss, rest_dens, window_dens, dens_* = False, ckpt_path -- omitted)."""

LIQUID3D = dict(  # configs/Liquid3d.yml:7-31
    name="SymNet",
    layer_channels=[[[8]], [[16], [8], [4]], [[32], [16], [8]], [[32]], [[3]]],
    kernel_size=[4, 4, 4], sym_kernel_size=[6, 6, 6], coordinate_mapping="ball_to_cube_volume_preserving",
    interpolation="linear", window="poly6", window_sym="peak", strides=[1, 2, 4], particle_radii=[0.1, 0.2, 0.4],
    timestep=0.02, grav=-9.81, out_scale=[0.0078125, 0.0078125, 0.0078125], centralize=True,
    voxel_size=[0.025, 0.025, 0.025], sym_axis=1, circular=False, add_merge=True, use_pre_adv=False, use_acc=False)

WATERRAMPS = dict(  # configs/WaterRamps.yml:7-31
    name="SymNet",
    layer_channels=[[[8]], [[16], [8], [4]], [[32], [16], [8]], [[32], [16], [8]], [[32]], [[2]]],
    kernel_size=[1, 8, 8], sym_kernel_size=[1, 8, 8], coordinate_mapping="ball_to_cube_volume_preserving",
    interpolation="linear", window="poly6", window_sym="peak", strides=[1, 2, 4], particle_radii=[0.02, 0.04, 0.08],
    timestep=0.0025, grav=-9.81, out_scale=[1.0e-4, 1.0e-4, 0.0], centralize=True, voxel_size=[0.01, 0.01, 0.0],
    sym_axis=1, circular=False, add_merge=True, use_pre_adv=False, use_acc=False)

WBC_SPH = dict(  # configs/WBC-SPH.yml:7-33
    name="SymNet",
    layer_channels=[[[8]], [[16], [8], [4], [4]], [[32], [16], [8], [4]], [[32], [16], [8], [4]], [[32]], [[2]]],
    kernel_size=[1, 8, 8], sym_kernel_size=[1, 8, 8], coordinate_mapping="ball_to_cube_volume_preserving",
    interpolation="linear", window="poly6", window_sym="peak", strides=[1, 2, 4, 8],
    particle_radii=[0.01, 0.02, 0.04, 0.08], timestep=0.0025, grav=-9.81, out_scale=[6.25e-06, 6.25e-06, 0.0],
    centralize=True, voxel_size=[0.005, 0.005, 0.0], sym_axis=1, circular=False, add_merge=True,
    use_pre_adv=False, transformation=dict(grav_eqvar=[0, -1, 0]))

COLUMN_HRNET = dict(  # configs/column/hrnet.yml:34-54
    name="HRNet",
    layer_channels=[[[8]], [[16], [8], [4], [4]], [[16], [8], [4], [4]], [[16]], [[1]]],
    kernel_size=[1, 8, 1], coordinate_mapping="ball_to_cube_volume_preserving", interpolation="linear",
    window="poly6", strides=[1, 2, 4, 8], particle_radii=[0.01, 0.02, 0.04, 0.08], timestep=0.0025, grav=-10.0,
    out_scale=[0.0, 6.25e-06, 0.0], centralize=True, voxel_size=[0.0, 0.005, 0.0], circular=False, add_merge=True,
    use_pre_adv=False)

CCONV2D = dict(  # configs/other/cconv.yml:7-21
    name="CConv", layer_channels=[32, 64, 64, 3], kernel_size=[1, 4, 4],
    coordinate_mapping="ball_to_cube_volume_preserving", interpolation="linear", window="poly6",
    ignore_query_points=True, use_bnds=False, particle_radii=[0.025], timestep=0.0025, grav=-9.81,
    out_scale=[6.25e-06, 6.25e-06, 0.0])

BY_NAME = {"Liquid3d": LIQUID3D, "WaterRamps": WATERRAMPS, "WBC-SPH": WBC_SPH, "column/hrnet": COLUMN_HRNET,
           "other/cconv": CCONV2D}
