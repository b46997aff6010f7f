"""The `prompt=` route of the two drop-in pipelines -- what both canonical callers use (reference app.py:708-719,
test_code/run_cogvideox_FrameIn_mass_evaluation.py:206-213) -- against runs of the REFERENCE pipelines' own `__call__`
with prompt STRINGS (tools/golden/make_golden.py: `encode_prompt` / `_get_t5_prompt_embeds` of
pipelines/pipeline_wan_i2v_motion_FrameINO.py:206-337 and pipelines/pipeline_cogvideox_i2v_motion_FrameINO.py:226-348,
unmodified).  The text encoder is the real `transformers` class at toy width on both sides, the tokenizer a character
stand-in (tests/text_stub.py); the fixture holds the encoder's weights, the strings, the embeddings and the latents.

CPU: the mirrors' encode_prompt reproduces the reference's embeddings (prompt_clean, zero padding past the true length for
Wan; truncation, no mask for CogVideoX).  GPU: `pipe(prompt="...", negative_prompt="")` end to end on the HIP path.
"""
import pytest
import torch

from tests.text_stub import CharTokenizer, tiny_text_encoder


def _text(a, kind, device="cpu"):
    sd = {k[3:]: v for k, v in a.items() if isinstance(k, str) and k.startswith("te/")}
    assert sd, "fixture without text-encoder weights: regenerate with tools/golden/make_golden.py"
    return CharTokenizer(), tiny_text_encoder(kind, sd).to(device)


def test_wan_encode_prompt_reproduces_the_reference_embeddings(golden):
    from frameino_amd.pipeline_wan_i2v_motion_frameino import WanImageToVideoPipeline, prompt_clean
    _, _, a = golden("wan_pipe_tiny")
    tok, te = _text(a, "umt5")
    pipe = WanImageToVideoPipeline(tokenizer=tok, text_encoder=te, expand_timesteps=True)
    prompt = str(a["prompt"])
    assert prompt_clean(prompt) == "A red ball rolls to the right & stops."      # double unescape + whitespace collapse
    pe, ne = pipe.encode_prompt(prompt, str(a["negative_prompt"]), True, 1, max_sequence_length=512, device="cpu")
    torch.testing.assert_close(pe, a["prompt_embeds_from_text"], atol=2e-5, rtol=1e-4)
    torch.testing.assert_close(ne, a["negative_embeds_from_text"], atol=2e-5, rtol=1e-4)
    n = len(prompt_clean(prompt)) + 1                                             # + EOS
    pe, ne = pe.detach(), ne.detach()
    assert float(pe[0, n:].abs().max()) == 0.0 and float(pe[0, :n].abs().min(dim=1).values.max()) > 0     # :235-238
    assert float(ne[0, 1:].abs().max()) == 0.0                                    # "" = EOS alone
    # list prompts, several videos per prompt, and the reference's type / batch checks (:313-325)
    pe2, ne2 = pipe.encode_prompt([prompt, "x"], None, True, 2, max_sequence_length=64, device="cpu")
    assert pe2.shape == (4, 64, 16) and ne2.shape == (4, 64, 16) and torch.equal(pe2[0], pe2[1])
    with pytest.raises(TypeError, match="same type"):            # (both are lists by then: only another container type trips it)
        pipe.encode_prompt(prompt, ("a",), True, 1, device="cpu")
    with pytest.raises(ValueError, match="batch size"):
        pipe.encode_prompt([prompt], ["a", "b"], True, 1, device="cpu")


def test_cog_encode_prompt_reproduces_the_reference_embeddings(golden):
    from frameino_amd.pipeline_cogvideox_i2v_motion_frameino import CogVideoXImageToVideoPipeline
    _, _, a = golden("cog_pipe_tiny")
    tok, te = _text(a, "t5")
    pipe = CogVideoXImageToVideoPipeline(tokenizer=tok, text_encoder=te)
    with pytest.warns(UserWarning, match="truncated"):                            # 10 characters + EOS into 8 slots (:247-253)
        pe, ne = pipe.encode_prompt(str(a["prompt"]), str(a["negative_prompt"]), True, 1, max_sequence_length=8,
                                    device="cpu")
    torch.testing.assert_close(pe, a["prompt_embeds_from_text"], atol=2e-5, rtol=1e-4)
    torch.testing.assert_close(ne, a["negative_embeds_from_text"], atol=2e-5, rtol=1e-4)
    # no guidance: no negative branch (:312); embeddings handed in are returned as they are
    pe3, ne3 = pipe.encode_prompt(None, None, False, 1, prompt_embeds=pe, device="cpu")
    assert pe3 is pe and ne3 is None
    with pytest.raises(TypeError, match="same type"):
        pipe.encode_prompt("a", ("b",), True, 1, max_sequence_length=8, device="cpu")
    # the reference's argument checks (:461-523)
    img = torch.zeros(1, 3, 64, 64)
    with pytest.raises(ValueError, match="Cannot forward both `prompt` and `prompt_embeds`"):
        pipe.check_inputs(img, "a", 64, 64, None, ["latents"], None, pe, None)
    with pytest.raises(ValueError, match="Provide either `prompt` or `prompt_embeds`"):
        pipe.check_inputs(img, None, 64, 64, None, ["latents"])
    with pytest.raises(ValueError, match="divisible by 8"):
        pipe.check_inputs(img, "a", 60, 64, None, ["latents"])
    with pytest.raises(ValueError, match="must have the same shape"):
        pipe.check_inputs(img, None, 64, 64, None, ["latents"], None, pe, ne[:, :4])
    # without a tokenizer / text encoder a prompt string is an error that says what to pass (never a silent skip)
    with pytest.raises(ValueError, match="tokenizer"):
        CogVideoXImageToVideoPipeline().encode_prompt("a", None, False, device="cpu")


@pytest.mark.gpu
def test_wan_call_with_prompt_strings_vs_the_reference_run(golden):
    import PIL.Image
    from frameino_amd.pipeline_wan_i2v_motion_frameino import WanImageToVideoPipeline
    from frameino_amd.schedulers import FlowMatchEulerDiscreteScheduler
    from tests.parity import hip_wan_model, record, rel_rms
    from tests.test_wan_vae_gpu import _vae
    cfg, sd, a = golden("wan_pipe_tiny")
    m = hip_wan_model(cfg, {k[4:]: v for k, v in sd.items() if k.startswith("dit.")}, "cuda")
    vae, _ = _vae(golden, "wan_pipe_tiny", prefix="vae")
    tok, te = _text(a, "umt5", "cuda")
    pipe = WanImageToVideoPipeline(tokenizer=tok, text_encoder=te, vae=vae,
                                   scheduler=FlowMatchEulerDiscreteScheduler(shift=5.0), transformer=m, expand_timesteps=True)
    img = PIL.Image.fromarray(a["image"].numpy())
    kw = dict(image=img, traj_tensor=a["traj"], ID_tensor=a["id_tensor"], height=img.size[1], width=img.size[0],
              num_frames=a["traj"].shape[0], num_inference_steps=int(a["steps"]), guidance_scale=float(a["guidance"]),
              output_type="latent")
    lat = pipe(prompt=str(a["prompt"]), negative_prompt=str(a["negative_prompt"]), latents=a["latents0"].clone(), **kw).frames
    r = rel_rms(lat, a["out_latents_prompt"])
    record("wan_call[prompt strings]", "rel_rms hip bf16 __call__(prompt=...) latents vs reference fp32 run", r, 6e-2)
    assert lat.shape == a["out_latents_prompt"].shape and r < 6e-2, r
    # the reference zero-pads the prompt to 512 rows (:235-238) and attends to all of them; this run folded the padding of both
    # prompts into one key each and re-associated the text out-projection -- against the REFERENCE's latents
    folded = [v[2] for v in m._text_cache.values()]
    assert folded and all(t.tail is not None and t.w2 is not None and t.lt == 128 for t in folded)
    m.dedup_text_padding = False
    lat_all_rows = pipe(prompt=str(a["prompt"]), negative_prompt=str(a["negative_prompt"]), latents=a["latents0"].clone(), **kw).frames
    m.dedup_text_padding = True
    assert all(v[2].tail is None for v in m._text_cache.values())
    r_all = rel_rms(lat_all_rows, a["out_latents_prompt"])
    record("wan_call[prompt strings, all 512 text rows]", "rel_rms hip bf16 latents vs reference fp32 run (padding NOT folded)", r_all, 6e-2)
    assert r < 1.2 * r_all + 2e-3, (r, r_all)
    # the text really went in: the run with the recorded random embeddings is a different clip
    assert rel_rms(a["out_latents_prompt"], a["out_latents"]) > 5 * r
    with pytest.raises(ValueError, match="Cannot forward both `prompt` and `prompt_embeds`"):
        pipe(prompt="a", prompt_embeds=a["prompt_embeds"], latents=a["latents0"].clone(), **kw)


@pytest.mark.gpu
def test_cog_call_with_prompt_strings_vs_the_reference_run(golden):
    import PIL.Image
    from tests.parity import record, rel_rms
    from tests.test_cog_model_gpu import _cog_pipe
    pipe, a, _ = _cog_pipe(golden, with_vae=True)
    pipe.tokenizer, pipe.text_encoder = _text(a, "t5", "cuda")
    H, W = a["image"].shape[:2]
    kw = dict(image=PIL.Image.fromarray(a["image"].numpy()), traj_tensor=a["traj"].to("cuda"), ID_tensor=a["id_tensor"].to("cuda"),
              height=H, width=W, num_frames=a["traj"].shape[0], num_inference_steps=int(a["steps"]),
              guidance_scale=float(a["guidance"]), add_ID_reference_augment_noise=False, max_sequence_length=8,
              output_type="latent")
    torch.manual_seed(7)
    with pytest.warns(UserWarning, match="truncated"):
        lat = pipe(prompt=str(a["prompt"]), negative_prompt=str(a["negative_prompt"]), latents=a["latents0"].to("cuda"), **kw).frames
    r = rel_rms(lat, a["out_ddim_prompt"])
    record("cog_call[prompt strings]", "rel_rms hip bf16 __call__(prompt=...) latents vs reference fp32 run", r, 8e-2)
    assert lat.shape == a["out_ddim_prompt"].shape and r < 8e-2, r
    assert rel_rms(a["out_ddim_prompt"], a["out_ddim"]) > 3 * r
    # guidance without a negative branch handed in: the empty negative prompt is encoded, as the reference does (:312-347)
    torch.manual_seed(7)
    with pytest.warns(UserWarning, match="truncated"):
        lat2 = pipe(prompt=str(a["prompt"]), latents=a["latents0"].to("cuda"), **kw).frames
    assert torch.equal(lat, lat2)
