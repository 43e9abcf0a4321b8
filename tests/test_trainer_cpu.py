"""Host logic of FusedAdam / FlatParams that needs no kernel: which parameters a step touches (torch.optim.Adam skips
parameters without a gradient), the optimizer state layout, and the flat-buffer link check."""
import pytest
import torch

import maavss_amd
from maavss_amd.trainer import FlatParams, FusedAdam

SHAPES = ([2, 2, 64, 129], [2, 1, 8, 128, 128], 8)


def _numel(shape):
    n = 1
    for d in shape:
        n *= d
    return n


def test_adam_runs_cover_exactly_the_touched_parameters():
    model = maavss_amd.AV_Fusion_Model_Frames(*SHAPES)
    opt = FusedAdam(model, lr=1e-3)
    f = opt.flat
    assert opt._runs() == []                                       # nothing received a gradient yet
    f.mark(model._param_names)                                     # what forward()'s backward produces
    runs = opt._runs()
    assert opt._runs() == runs and all(v == 0 for v in opt.steps.values())      # planning is pure: nothing committed yet
    opt._commit(runs)
    covered = sum(hi - lo for lo, hi, _, _ in runs)
    want = sum((_numel(f.shapes[n]) + 63) // 64 * 64 for n in model._param_names)
    assert covered == want and all(st == 1 for _, _, st, _ in runs)
    # stft_decoder.* (no gradient under forward(), avse_model_final.py:258-274) is outside every run
    for n in f.names:
        inside = any(lo <= f.offsets[n] < hi for lo, hi, _, _ in runs)
        assert inside == (not n.startswith("stft_decoder.")), n
    sd = opt.state_dict()
    stepped = {f.torch_order[i] for i in sd["state"]}
    assert stepped == set(model._param_names)                      # never-stepped parameters have no state entry
    assert all(float(v["step"]) == 1.0 for v in sd["state"].values())
    # frozen sub-network: requires_grad False keeps a parameter out even if a stale mark exists
    opt.zero_grad()
    assert f.touched == set()
    model.toggle_enc_grads(False)
    f.mark(model._param_names)
    runs = opt._runs()
    opt._commit(runs)
    for n in f.names:
        inside = any(lo <= f.offsets[n] < hi for lo, hi, _, _ in runs)
        frozen = n.startswith("visual_encoder.") or n.startswith("stft_encoder.") or n.startswith("stft_decoder.")
        assert inside == (not frozen), n
    assert opt.steps["fc1.weight"] == 2 and opt.steps["visual_encoder.0.weight"] == 1
    # runs with different step counts are never merged
    assert len({st for _, _, st, _ in runs}) == 1


def test_state_dict_round_trip_keeps_per_parameter_steps():
    model = maavss_amd.AV_Fusion_Model_Frames(*SHAPES)
    opt = FusedAdam(model, lr=2e-4)
    opt.flat.mark(["fc1.weight", "fc2.weight"])
    opt._commit(opt._runs())
    opt._commit(opt._runs())
    opt.flat.mark(["a_fc1.0.weight"])
    opt._commit(opt._runs())
    sd = opt.state_dict()
    opt2 = FusedAdam(maavss_amd.AV_Fusion_Model_Frames(*SHAPES), lr=1.0)
    opt2.load_state_dict(sd)
    assert opt2.steps == opt.steps and opt2.lr == 2e-4
    assert opt2.steps["fc1.weight"] == 3 and opt2.steps["a_fc1.0.weight"] == 1 and opt2.steps["lstm.weight_ih_l0"] == 0


def test_link_check_repairs_grads_and_refuses_moved_parameters():
    model = maavss_amd.AV_Fusion_Model_Frames(*SHAPES)
    flat = FlatParams(model)
    flat.check_links()
    model.zero_grad()                                              # torch default set_to_none=True detaches every .grad
    assert model.fc2.weight.grad is None
    flat.mark(["fc2.weight"])
    flat.check_links()
    assert model.fc2.weight.grad.data_ptr() == flat.grad_views["fc2.weight"].data_ptr()
    assert "fc2.weight" not in flat.touched                        # grad None == no gradient this step
    model.fc2.weight.grad = torch.ones_like(model.fc2.weight)      # a foreign gradient tensor is folded back in
    flat.check_links()
    assert float(flat.grad_views["fc2.weight"].sum()) == model.fc2.weight.numel() and "fc2.weight" in flat.touched
    model.double()                                                 # re-homes p.data: cannot be repaired silently
    with pytest.raises(maavss_amd._lib.MaavssError):
        flat.check_links()
