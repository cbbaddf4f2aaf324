"""sbsim_amd -- MI355X-native batched building-thermal step behind sbsim's Environment API.

    from sbsim_amd.environment import BatchedEnvironment, GymVectorEnv, SimConfig
    from sbsim_amd.floorplan import FloorPlan, Materials

`sbsim_amd.environment`   BatchedEnvironment / BatchedSimulator (ctypes over include/sbsim_amd.h)
`sbsim_amd.floorplan`     floor-plan preprocessing -> per-cell class tables
`sbsim_amd.host_inputs`   weather, schedule, occupancy, tariffs, time features (host side)
`sbsim_amd.distributed`   one process per GPU: shard arithmetic, end-of-rollout return gather
`sbsim_amd.build`         hipcc (gfx950) build of libsbsim_amd.so

There is no CPU fallback: without the HIP library or a GPU every constructor raises.
"""
__version__ = "0.1.0"
