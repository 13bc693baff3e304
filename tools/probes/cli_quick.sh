cd $GRAFT_REPO_ROOT
run() { echo "== $*"; timeout 300 python -m cmcd_amd.main "$@" --config.iters 150 --config.mfvi_iters 100 --config.n_samples 100 --config.n_input_dist_seeds 3 2>&1 | grep -E "Error|error|Done training, got|not implemented|Traceback" | tail -3; }
run --config.model gmm --config.boundmode MCD_CAIS_sn --config.nn_arch geffner --config.N 64 --config.nbridges 6 --config.init_eps 0.01 --config.lr 1e-3
run --config.model many_gmm --config.boundmode MCD_CAIS_var_sn --config.nn_arch dds --config.N 64 --config.nbridges 6 --config.init_eps 0.01 --config.lr 1e-3
run --config.model funnel --config.boundmode MCD_ULA_sn --config.nn_arch geffner --config.N 64 --config.nbridges 6 --config.init_eps 0.01 --config.lr 1e-3
run --config.model gmm --config.boundmode MCD_ULA --config.N 64 --config.nbridges 6 --config.init_eps 0.01 --config.lr 1e-3
run --config.model many_gmm --config.boundmode MCD_CAIS_UHA_sn --config.nn_arch dds --config.N 64 --config.nbridges 6 --config.lr 1e-3
run --config.model lgcp --config.boundmode MCD_CAIS_sn --config.N 20 --config.nbridges 4 --config.lr 1e-4
run --config.model lgcp --config.boundmode MCD_CAIS_var_sn --config.N 20 --config.nbridges 4 --config.lr 1e-4
run --config.model lgcp --config.boundmode MCD_CAIS_UHA_sn --config.N 20 --config.nbridges 4 --config.lr 1e-4
run --config.model lgcp --config.boundmode MCD_ULA --config.N 40 --config.nbridges 4 --config.lr 1e-4
