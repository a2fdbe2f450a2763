set -e
cd /root/repo
D=/tmp/ds; rm -rf $D /tmp/exp
python mega-nerf_amd/tools/make_synthetic_dataset.py --out $D --images 12 --val_every 6 --size 64 --samples 64 128 > /dev/null
cd mega-nerf_amd
python -m mega_nerf.train --dataset_path $D --exp_name /tmp/exp --coarse_samples 64 --fine_samples 128 --near 0.01 --ray_altitude_range -0.5 0.2 --val_scale_factor 1 --batch_size 1024 --train_iterations 1500 --ckpt_interval 500 --val_interval 500 2>&1 | grep -E "iter (100|500|1000|1500):|Average" | tail -8
