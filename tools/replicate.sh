# The reference README's "replicate" command lines (README.md:53,63,73), flag for flag, through `python -m cmcd_amd.main`,
# once per number of bridges of the result tables kept in src/notebooks/plotting_rebuttal.ipynb.
# usage (GPU box): bash tools/replicate.sh "8 16 32 64 128 256" "8 64 256" "8 16"   -> gpurun_out/replicate.txt
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
out=gpurun_out/replicate.txt
: > $out
run() { echo "== $*" >> $out; t0=$SECONDS; timeout 3000 python -m cmcd_amd.main "$@" 2>&1 | grep -E "Done training|iterations in|W2|Error|error|Implemented" >> $out; echo "wall $((SECONDS - t0)) s" >> $out; }
for k in $1; do
run --config.boundmode MCD_CAIS_sn --config.model funnel --config.N 300 --config.alpha 0.05 --config.emb_dim 48 --config.init_eps 0.1 -config.init_sigma 1 --config.iters 11000 --noconfig.pretrain_mfvi --config.train_vi --noconfig.train_eps --config.wandb.name "funnel replicate w/ cos_sq" --config.lr 0.01 --config.n_samples 2000 --config.eps_schedule cos_sq --config.nbridges $k
done
for k in $2; do
run --config.boundmode MCD_CAIS_sn --config.model gmm --config.N 300 --config.alpha 0.05 --config.emb_dim 20 --config.init_eps 0.01 -config.init_sigma 1 --config.iters 11000 --noconfig.pretrain_mfvi --config.train_vi --noconfig.train_eps --config.wandb.name "gmm replicate" --config.lr 0.001 --config.n_samples 500 --config.nbridges $k
done
for k in $3; do
run --config.boundmode MCD_CAIS_sn --config.model lgcp --config.N 20 --config.alpha 0.05 --config.emb_dim 20 --config.init_eps 0.00001 -config.init_sigma 1 --config.iters 37500 --config.pretrain_mfvi --config.train_vi --config.train_eps --config.wandb.name "lgcp replicate" --config.lr 0.0001 --config.n_samples 500 --config.mfvi_iters 20000 --config.nbridges $k
done
cat $out
