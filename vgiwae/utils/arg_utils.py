def construct_experiment_subdir(args):
    """`<experiment_subdir_base>/seed_m<model seed>_d<data seed>`: the run directory the entry scripts log under
    (reference: vgiwae/utils/arg_utils.py:1-2)."""
    seeds = "seed_m{}_d{}".format(args.seed_everything, args.data.setup_seed)
    return "/".join((str(args.experiment_subdir_base), seeds))
