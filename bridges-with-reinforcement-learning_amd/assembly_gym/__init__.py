"""assembly_gym drop-in: same import surface as the reference package
(assembly_gym/assembly_gym in syghmon/bridges-with-reinforcement-learning), with the simulation
(placement, contact interfaces, RBE stability, rasters) executed by the HIP kernels of
libbridges_hip.so.  There is no CPU implementation behind it."""
