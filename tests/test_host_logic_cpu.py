"""Host-side contracts that need no GPU: config precedence (config/parse_args.py semantics), optimiser-state layout checks, the slab's
frozen set.  """
import importlib

import pytest
import torch


def test_cli_overrides_yaml_for_append_options(pkg):
    """configargparse: the command line overrides the file -- also for action='append' options, where argparse alone would append."""
    cfg = importlib.import_module(pkg.__name__ + ".config")
    path = cfg.resolve("multimodal_context")
    base = cfg.parse_args(["-c", path])
    assert len(base.mean_dir_vec) == 1 and len(base.mean_dir_vec[0]) == 27 and len(base.train_data_path) == 1
    over = cfg.parse_args(["-c", path, "--train_data_path", "/elsewhere", "--mean_dir_vec", "1", "2", "3", "--learning_rate", "0.5"])
    assert over.train_data_path == ["/elsewhere"]                    # replaced, not appended to the file's value
    assert over.mean_dir_vec == [[1.0, 2.0, 3.0]]
    assert over.learning_rate == 0.5 and over.hidden_size == base.hidden_size and over.val_data_path == base.val_data_path
    eq = cfg.parse_args(["-c", path, "--train_data_path=/with_equals"])
    assert eq.train_data_path == ["/with_equals"]


class _Net(torch.nn.Module):
    def __init__(self, freeze):
        super().__init__()
        self.a = torch.nn.Linear(4, 4)
        self.emb = torch.nn.Embedding(6, 4)
        self.b = torch.nn.Linear(4, 2)
        self.emb.weight.requires_grad_(not freeze)


def test_adam_state_refuses_a_different_slab_layout(pkg):
    params = importlib.import_module(pkg.__name__ + ".params")
    optim = importlib.import_module(pkg.__name__ + ".optim")

    class Eng:
        def __init__(self, freeze):
            self.slab = params.ParamSlab(_Net(freeze))
    e0, e1 = Eng(False), Eng(True)
    assert e0.slab.names != e1.slab.names and e1.slab.names[-1] == "emb.weight" and e1.slab.n_train < e1.slab.numel
    o0, o1 = optim.FusedAdam(e0, lr=1e-3), optim.FusedAdam(e1, lr=1e-3)
    sd = o0.state_dict()
    optim.FusedAdam(Eng(False), lr=1e-3).load_state_dict(sd)          # same layout: accepted
    with pytest.raises(ValueError, match="different parameter layout"):
        o1.load_state_dict(sd)


def test_slab_detects_requires_grad_changed_after_layout(pkg):
    params = importlib.import_module(pkg.__name__ + ".params")
    net = _Net(False)
    slab = params.ParamSlab(net)
    slab.ensure()
    net.emb.weight.requires_grad_(False)
    with pytest.raises(RuntimeError, match="requires_grad"):
        slab.ensure()
