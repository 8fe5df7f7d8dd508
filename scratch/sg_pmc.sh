export RAYS_HIP_LIB=$PWD/rays_amd/lib/librays_hip_exp_tier.so
bash scratch/pmc_mix.sh --config $PWD/configs/cfg5_axisym256k_sg_damp.in --steps 1 --warmup 1
