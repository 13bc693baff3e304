# The reference README's 40-mode GMM command lines (README.md:26,30,34; W&B-only results upstream: the known answer is ln Z = 0)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
out=gpurun_out/replicate_many_gmm.txt
: > $out
run() { echo "== $*" >> $out; t0=$SECONDS; timeout 3000 python -m cmcd_amd.main "$@" 2>&1 | grep -E "Done training|iterations in|W2|Error|error|Implemented" >> $out; echo "wall $((SECONDS - t0)) s" >> $out; }
run --config.model many_gmm --config.boundmode MCD_CAIS_sn --config.N 2000 --config.nbridges 256 --noconfig.pretrain_mfvi --config.init_sigma 60 --config.grad_clipping --config.init_eps 1 --config.eps_schedule cos_sq --config.lr 0.001 --noconfig.train_eps --noconfig.train_vi --config.wandb.name "kl 40gmm pis net eps=1, cos_sq" --config.nn_arch dds
run --config.model many_gmm --config.boundmode MCD_CAIS_var_sn --config.N 2000 --config.nbridges 256 --noconfig.pretrain_mfvi --config.init_sigma 15 --config.grad_clipping --config.init_eps 0.65 --config.emb_dim 130 --config.lr 0.005 --noconfig.train_eps --noconfig.train_vi --config.wandb.name "logvar 40gmm"
run --config.model many_gmm --config.boundmode MCD_CAIS_sn --config.N 2000 --config.nbridges 256 --noconfig.pretrain_mfvi --config.init_sigma 15 --config.grad_clipping --config.init_eps 0.1 --config.emb_dim 130 --config.lr 0.005 --noconfig.train_eps --noconfig.train_vi --config.wandb.name "kl 40gmm"
cat $out
