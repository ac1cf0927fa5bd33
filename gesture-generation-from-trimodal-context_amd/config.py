"""Configuration: the reference's config/parse_args.py (configargparse over a YAML file) on plain argparse + PyYAML.

Same option names, types and defaults as parse_args.py:16-66; values come, in rising priority, from the defaults, the YAML file
given with -c/--config, the command line.  `load_config(name_or_path, **overrides)` is the programmatic form used by bench.py, the
examples and the tests (one source of hyper-parameters instead of hand-copied Namespaces); the two YAML files of the hot path ship in
this package's config/ directory.
"""
import argparse
import os

import yaml

CONFIG_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "config")


def str2bool(v):
    if isinstance(v, bool):
        return v
    if v.lower() in ("yes", "true", "t", "y", "1"):
        return True
    if v.lower() in ("no", "false", "f", "n", "0"):
        return False
    raise argparse.ArgumentTypeError("Boolean value expected.")


def _parser():
    p = argparse.ArgumentParser()
    p.add_argument("-c", "--config", required=True, help="Config file path")
    p.add_argument("--name", type=str, default="main")
    p.add_argument("--train_data_path", action="append")
    p.add_argument("--val_data_path", action="append")
    p.add_argument("--test_data_path", action="append")
    p.add_argument("--model_save_path")
    p.add_argument("--pose_representation", type=str, default="3d_vec")
    p.add_argument("--mean_dir_vec", action="append", type=float, nargs="*")
    p.add_argument("--mean_pose", action="append", type=float, nargs="*")
    p.add_argument("--random_seed", type=int, default=-1)
    p.add_argument("--save_result_video", type=str2bool, default=True)
    # word embedding
    p.add_argument("--wordembed_path", type=str, default=None)
    p.add_argument("--wordembed_dim", type=int, default=100)
    p.add_argument("--freeze_wordembed", type=str2bool, default=False)
    # model
    p.add_argument("--model", type=str)
    p.add_argument("--epochs", type=int, default=10)
    p.add_argument("--batch_size", type=int, default=50)
    p.add_argument("--dropout_prob", type=float, default=0.3)
    p.add_argument("--n_layers", type=int, default=2)
    p.add_argument("--hidden_size", type=int, default=200)
    p.add_argument("--z_type", type=str, default="none")
    p.add_argument("--input_context", type=str, default="both")
    # dataset
    p.add_argument("--motion_resampling_framerate", type=int, default=24)
    p.add_argument("--n_poses", type=int, default=50)
    p.add_argument("--n_pre_poses", type=int, default=5)
    p.add_argument("--subdivision_stride", type=int, default=5)
    p.add_argument("--loader_workers", type=int, default=0)
    # GAN parameter
    p.add_argument("--GAN_noise_size", type=int, default=0)
    # training
    p.add_argument("--learning_rate", type=float, default=0.001)
    p.add_argument("--discriminator_lr_weight", type=float, default=0.2)
    p.add_argument("--loss_regression_weight", type=float, default=50)
    p.add_argument("--loss_gan_weight", type=float, default=1.0)
    p.add_argument("--loss_kld_weight", type=float, default=0.1)
    p.add_argument("--loss_reg_weight", type=float, default=0.01)
    p.add_argument("--loss_warmup", type=int, default=-1)
    # eval
    p.add_argument("--eval_net_path", type=str, default="")
    return p


def _yaml_to_argv(cfg):
    """A config-file entry behaves like the same option on the command line (configargparse semantics): a YAML list becomes the
    option's nargs / append values, so `mean_dir_vec: [..27 floats..]` parses to [[..27 floats..]] exactly as in the reference."""
    argv = []
    for key, val in cfg.items():
        if val is None:
            continue
        argv.append("--" + key)
        argv.extend(str(v) for v in val) if isinstance(val, (list, tuple)) else argv.append(str(val))
    return argv


def parse_args(argv=None):
    """parse_args.py:16-66.  argv defaults to sys.argv[1:]."""
    p = _parser()
    pre, _ = p.parse_known_args(argv)
    with open(pre.config) as f:
        cfg = yaml.safe_load(f) or {}
    known = {a.dest for a in p._actions}
    unknown = sorted(set(cfg) - known)
    if unknown:
        raise SystemExit(f"{pre.config}: unknown option(s) {unknown}")
    import sys
    cli = list(sys.argv[1:] if argv is None else argv)
    # configargparse semantics: the command line overrides the file.  For plain options a later occurrence wins by itself; an
    # action='append' option (train_data_path, mean_dir_vec ...) would instead be appended to the file's value, so a config key is
    # dropped when the command line names the same option
    on_cli = set()
    for tok in cli:
        if tok.startswith("-"):
            act = p._option_string_actions.get(tok.split("=", 1)[0])
            if act is not None:
                on_cli.add(act.dest)
    cfg = {k: v for k, v in cfg.items() if k not in on_cli}
    args = p.parse_args(_yaml_to_argv(cfg) + cli)
    if args.model is None or args.model_save_path is None:
        raise SystemExit("--model and --model_save_path are required (config file or command line)")
    return args


def resolve(name_or_path):
    if os.path.isfile(name_or_path):
        return name_or_path
    path = os.path.join(CONFIG_DIR, name_or_path if name_or_path.endswith(".yml") else name_or_path + ".yml")
    if not os.path.isfile(path):
        raise FileNotFoundError(f"no config {name_or_path!r} (looked in {CONFIG_DIR})")
    return path


def load_config(name_or_path="multimodal_context", **overrides):
    """The Namespace parse_args would return for `-c <file>`, with keyword overrides applied on top (any attribute, also ones
    parse_args does not know, e.g. pose_dim)."""
    args = parse_args(["-c", resolve(name_or_path)])
    for k, v in overrides.items():
        setattr(args, k, v)
    return args
