def construct_experiment_subdir(args):
    """Experiment directory name used by the entry scripts (reference: vgiwae/utils/arg_utils.py:1-2)."""
    return "{}/seed_m{}_d{}".format(args.experiment_subdir_base, args.seed_everything, args.data.setup_seed)
