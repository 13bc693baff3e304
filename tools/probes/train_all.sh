# end-to-end training runs of the BASELINE configurations through tools/train.py (true ln Z = 0 for the synthetic targets)
cd $GRAFT_REPO_ROOT
run() { echo "== $*"; timeout 600 python tools/train.py "$@" 2>&1 | grep -E "Done training initial|before|iterations|after"; }
run --model gmm --boundmode MCD_CAIS_sn --N 300 --nbridges 8 --nn_arch geffner --emb_dim 20 --init_eps 0.01 --pretrain_mfvi --mfvi_iters 2000 --iters 10000 --lr 1e-3 --train_eps --train_vi --train_betas
run --model funnel --boundmode MCD_CAIS_sn --N 300 --nbridges 64 --nn_arch geffner --emb_dim 48 --init_eps 0.1 --eps_schedule cos_sq --pretrain_mfvi --mfvi_iters 2000 --iters 8000 --lr 1e-3 --train_eps --train_vi --train_betas
run --model many_gmm --boundmode MCD_CAIS_sn --N 2000 --nbridges 256 --nn_arch dds --init_sigma 60 --init_eps 1.0 --eps_schedule cos_sq --grad_clipping --iters 8000 --lr 1e-3 --train_eps --train_vi --train_betas
run --model many_gmm --boundmode MCD_CAIS_var_sn --N 2000 --nbridges 256 --nn_arch geffner --emb_dim 130 --init_sigma 15 --init_eps 0.65 --grad_clipping --iters 1500 --lr 1e-3
run --model many_gmm --boundmode MCD_ULA_sn --N 2000 --nbridges 64 --nn_arch geffner --emb_dim 20 --init_sigma 15 --init_eps 0.3 --iters 3000 --lr 1e-3 --train_eps --train_betas
run --model many_gmm --boundmode MCD_ULA --N 2000 --nbridges 64 --init_sigma 15 --init_eps 0.3 --iters 3000 --lr 1e-3 --train_eps --train_betas
