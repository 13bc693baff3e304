# smoke matrix of the reference-style command line: every (model, boundmode, nn_arch) a user can pick in the overdamped family
cd $GRAFT_REPO_ROOT
run() { echo "== $*"; timeout 300 python -m cmcd_amd.main "$@" --config.iters 200 --config.mfvi_iters 200 --config.n_samples 100 --config.n_input_dist_seeds 5 2>&1 | grep -E "Error|error|Done training, got|not implemented" | tail -3; }
for model in gmm funnel many_gmm; do
  for mode in MCD_CAIS_sn MCD_CAIS_var_sn MCD_ULA_sn MCD_ULA; do
    for arch in geffner dds; do
      run --config.model $model --config.boundmode $mode --config.nn_arch $arch --config.N 64 --config.nbridges 6 --config.init_eps 0.01 --config.lr 1e-3
    done
  done
done
run --config.model lgcp --config.boundmode MCD_CAIS_sn --config.N 8 --config.nbridges 4 --config.lr 1e-4
run --config.model many_gmm --config.boundmode MCD_CAIS_sn --config.N 64 --config.nbridges 6 --config.emb_dim 100 --config.lr 1e-3
run --config.model gmm --config.boundmode UHA --config.N 64
