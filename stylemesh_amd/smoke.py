"""One small invocation of the hot path on cuda:0, checked against the oracle (called by __graft_entry__.smoke()).
The oracle import below is the CHECKER; the product path never touches it."""
import numpy as np
import torch

from .data import synthetic as S


def run():
    import stylemesh_oracle as O   # oracle/ is put on sys.path by __graft_entry__.smoke()
    from .runtime.engine import EngineConfig, StepEngine
    assert torch.cuda.is_available(), "smoke() needs cuda:0"
    torch.cuda.set_device(0)
    level_hw = [(40, 56), (64, 88)]
    batch = S.make_view(3, view_hw=(40, 56), level_hw=level_hw, level_heights=[40, 64], min_pyramid_depth=0.9,
                        room=S.BoxRoom((6.0, 4.5, 2.8)))
    lw = {"content": 7e1, "style": 1e-4, "tex_reg": 5e3}
    sw = [1000., 1000., 10., 10., 1000.]
    vgg = S.seeded_vgg_state(7)
    style = S.style_image(43, 300, 270)
    rng = np.random.default_rng(9)
    init = [torch.from_numpy(((S.smooth_noise(rng, 3, 64 >> i, 64 >> i, cells=4) - 0.5) * 60 / (i + 1)).astype(np.float32))
            for i in range(4)]
    eng = StepEngine(EngineConfig(tex_w=64, tex_h=64, style_weights=sw, angle_threshold=30, style_pyramid_mode="multi",
                                  loss_weights=dict(lw), learning_rate=1.0, decay_step_size=3), vgg)
    eng.load_texture(init)
    eng.set_style_image(style)
    pipe = O.OraclePipeline(vgg, style, O.OracleConfig(style_weights=sw, angle_threshold=30, style_pyramid_mode="multi",
                                                       loss_weights=dict(lw), learning_rate=1.0, decay_step_size=3),
                            (64, 64), init_layers=init)
    eng.set_view(batch)
    lt = eng.loss_tensors()
    eng.forward_backward()
    mine = eng.losses(lt)
    ref_losses, ref_grads = pipe.grads(batch)
    for k in ("content", "style", "tex_reg", "total"):
        np.testing.assert_allclose(mine[k], float(ref_losses[k]), rtol=2e-4, err_msg=k)
    for i, (g, c, p) in enumerate(zip(eng.grads, eng.reg_coef, eng.layers)):
        ref = ref_grads[i]
        err = ((g + c * p).cpu() - ref).abs()
        mx = float(ref.abs().max())
        assert float((err > 1e-3 * ref.abs() + 2e-4 * mx).float().mean()) <= 0.03 and float(err.max()) <= 2e-2 * mx, \
            f"layer {i}: max err {float(err.max()):.3e} vs max|g| {mx:.3e}"
    eng.optimizer_step()
    pipe.apply_adam(ref_grads)
    for i in range(4):
        err = (eng.layers[i].cpu() - pipe.layers[i].detach().clamp(O.CLAMP_LO, O.CLAMP_HI)).abs()
        assert int((err > 1e-4).sum()) <= 3, f"layer {i}: {int((err > 1e-4).sum())} texels differ after one Adam step"
    print(f"smoke OK: losses {mine}")
