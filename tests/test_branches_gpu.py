"""Product branches that had no test and no oracle leg (VERDICT r3 "missing" #3), HIP path vs the CPU oracle / the reference's goldens:
`use_temp_objectness=False` (GROVE.py:183-195, 282-289, 313-317, 383-408, 437-448; mask_decoder.py:83-87, 200-205), `token_embeddings=`
(llava_with_region_arch.py:134-137; infer_iground.py:193), `grad_accumulation_steps > 1` (train.py:467, 739-782), the
`--train_mask_decoder` choice of the freeze policy (train.py:279-288) and a vocabulary that is not a multiple of 8 (the real one:
32000 + the added tokens, train.py:132-152, 330)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")
bf = torch.bfloat16


def to_dev(batch, dev, **extra):
    kw = batch.as_kwargs(**extra)
    for k in ("global_enc_images", "grounding_enc_images"):
        kw[k] = kw[k].to(dev).to(bf)
    for k in ("input_ids", "labels", "attention_masks", "offset"):
        kw[k] = kw[k].to(dev)
    return kw


def oracle_kwargs(batch, **extra):
    kw = batch.as_kwargs(**extra)
    for k in ("global_enc_images", "grounding_enc_images"):
        kw[k] = kw[k].to(bf).float()
    return kw


def test_use_temp_objectness_false_inference_and_training(dev):
    """Inference: every box is kept, `logits_temp_objectness` is None, `evaluate` returns a 2-tuple; training: four loss keys with the
    shipped weights (1, 2, 2), box L1 / loss terms / gradients against the oracle run in the same mode, and against the golden the
    imported reference produced without an objectness head (tests/golden/tiny_no_objectness_seed7.npz)."""
    from grove_amd import GROVEForCausalLM
    from grove_amd.model.GROVE import trainable_names
    from grove_amd.synthetic import TINY, synthetic_batch, synthetic_state_dict
    from oracle import grove_oracle as O
    d = TINY
    g = np.load(os.path.join(G, "tiny_no_objectness_seed7.npz"))
    w = tuple(float(x) for x in g["loss_weights"])
    sd = synthetic_state_dict(d)
    sd_r = {k: v.to(bf).float() for k, v in sd.items()}
    # ---- inference
    model = GROVEForCausalLM(dims=d, device=dev, state_dict=sd, det_token_idx=d.det_token_idx, num_frames=8, pe_dtype=torch.float32,
                             use_temp_objectness=False)
    ib = synthetic_batch(d, B=2, T=8, L=40, n_det=3, seed=8)
    out = model(**to_dev(ib, dev, inference=True))
    assert out["logits_temp_objectness"] is None
    counts = np.array([[x.shape[0] for x in l_] for l_ in out["pred_bboxes"]])
    assert (counts == g["infer/pred_bboxes_counts"]).all()
    with torch.no_grad():
        ref = O.model_forward(sd_r, d, **oracle_kwargs(ib, inference=True), use_temp_objectness=False)
    l1 = (out["flat_boxes"].cpu() - ref["flat_boxes"]).abs().mean().item()
    assert l1 < 1e-3, f"box L1 {l1}"
    px = torch.cat([x.reshape(-1) for l_ in out["pred_bboxes"] for x in l_]).float().cpu()
    rpx = torch.cat([x.reshape(-1) for l_ in ref["pred_bboxes"] for x in l_])
    assert (px - rpx).abs().max().item() < 640 * 4e-3
    assert np.abs(px.numpy() - g["infer/pred_bboxes"]).max() < 640 * 1.5e-2   # the reference's fp32-weight golden (bf16 weight rounding on top)
    # evaluate(): two values (GROVE.py:445-448)
    prompt = ib.input_ids[:, :14].to(dev)
    feats, fouts = model(mode="encode_images", images=ib.global_enc_images.to(dev).to(bf))
    emb = model(mode="get_grounding_encoder_embs", images=ib.grounding_enc_images.to(dev).to(bf))
    res = model(mode="evaluate", image_features=feats, image_forward_outs=fouts, images_dtype=bf, image_embeddings=emb, input_ids=prompt,
                original_size_list=ib.original_size_list, max_tokens_new=6, dense_pe=model(mode="get_dense_pe"), device=dev)
    assert isinstance(res, tuple) and len(res) == 2
    # ---- training
    names = trainable_names(d)
    tm = GROVEForCausalLM(dims=d, device=dev, state_dict=sd, det_token_idx=d.det_token_idx, num_frames=8, pe_dtype=torch.float32, train=True,
                          use_temp_objectness=False, ce_loss_weight=w[0], giou_loss_weight=w[1], temp_objectness_loss_weight=w[2])
    tb = synthetic_batch(d, B=2, T=8, L=48, n_det=2, seed=7, ragged=True)
    tm.zero_grad()
    tout = tm(**to_dev(tb, dev))
    assert sorted(k for k in tout if k.endswith("loss")) == ["ce_loss", "giou_loss", "l1_loss", "loss"]
    tm.backward(tout["loss"])
    torch.cuda.synchronize()
    sdg = {k: v.clone().requires_grad_(k in names) for k, v in sd_r.items()}
    tref = O.model_forward(sdg, d, **oracle_kwargs(tb), use_temp_objectness=False, loss_weights=w)
    tref["loss"].backward()
    for k in ("ce_loss", "giou_loss", "l1_loss", "loss"):
        a, b = float(tout[k]), float(tref[k])
        assert abs(a - b) <= 2e-2 * max(1.0, abs(b)), f"{k}: {a} vs {b}"
        assert abs(a - float(g["train/" + k])) <= 3e-2 * max(1.0, abs(float(g["train/" + k]))), k
    bad = []
    # the head is not a parameter of this model (GROVE.py:118-126 only creates it when the flag is on): no gradient slot, no key
    assert set(tm.trainable) == set(trainable_names(d, True, False)) == {n for n in names if "temporal_objectness_head" not in n}
    assert not any("temporal_objectness_head" in k for k in tm.state_dict())
    for n in tm.trainable:
        gr = tm._grad[n].detach().float().cpu()
        r = sdg[n].grad
        if n.endswith("conv3d.weight"):
            gr = gr.view(r.shape[0], 3, 3, 3, r.shape[1]).permute(0, 4, 1, 2, 3)
        gr, r = gr.reshape(-1), r.reshape(-1)
        if float(r.norm()) < 1e-7:
            continue
        cos = float(torch.nn.functional.cosine_similarity(gr, r, dim=0))
        ratio = float(gr.norm() / r.norm())
        if cos < 0.98 or not 0.9 < ratio < 1.1:
            bad.append((n, cos, ratio))
    assert not bad, bad[:5]


def test_token_embeddings_table_is_the_embedding_path_bit_for_bit(dev):
    """embed_tokens.py dumps `model.embed_tokens.weight` to a file and the inference drivers index that tensor instead of calling the
    embedding (llava_with_region_arch.py:134-137; infer_iground.py:193). With the dumped table every output — teacher-forced
    model_forward, the cached LM step, generate / evaluate — must equal the embedding path BIT FOR BIT; a table with one row changed
    must change the result (i.e. the argument is really read)."""
    from grove_amd import GROVEForCausalLM
    from grove_amd.synthetic import TINY, synthetic_batch, synthetic_state_dict
    d = TINY
    sd = synthetic_state_dict(d)
    model = GROVEForCausalLM(dims=d, device=dev, state_dict=sd, det_token_idx=d.det_token_idx, num_frames=8, pe_dtype=torch.float32)
    table = model.state_dict()["model.embed_tokens.weight"].clone()  # "the dump"
    ib = synthetic_batch(d, B=2, T=8, L=24, n_det=1, seed=3)
    prompt = ib.input_ids[:, :14].to(dev)
    feats, fouts = model(mode="encode_images", images=ib.global_enc_images.to(dev).to(bf))
    emb = model(mode="get_grounding_encoder_embs", images=ib.grounding_enc_images.to(dev).to(bf))
    pe = model(mode="get_dense_pe")

    def run(tab):
        return model(mode="evaluate", image_features=feats, image_forward_outs=fouts, images_dtype=bf, image_embeddings=emb, input_ids=prompt,
                     original_size_list=ib.original_size_list, max_tokens_new=10, token_embeddings=tab, dense_pe=pe, device=dev)
    ids0, boxes0, logits0 = run(None)
    ids1, boxes1, logits1 = run(table)
    assert torch.equal(ids0, ids1)
    for a, b in zip(sum(boxes0, []) + sum(logits0, []), sum(boxes1, []) + sum(logits1, [])):
        assert torch.equal(a, b)
    g0 = model.generate(input_ids=prompt, image_features=feats, max_new_tokens=6, output_hidden_states=True, return_dict_in_generate=True)
    g1 = model.generate(input_ids=prompt, image_features=feats, max_new_tokens=6, output_hidden_states=True, return_dict_in_generate=True,
                        token_embeddings=table)
    assert torch.equal(g0.sequences, g1.sequences) and all(torch.equal(a, b) for a, b in zip(g0.hidden_states, g1.hidden_states))
    # uncached LM forward (forward(past_key_values=None, ...) -> lm_forward) and one cached step
    lm0 = model(past_key_values=None, input_ids=prompt, image_features=feats, output_hidden_states=True)
    lm1 = model(past_key_values=None, input_ids=prompt, image_features=feats, output_hidden_states=True, token_embeddings=table)
    assert torch.equal(lm0.logits, lm1.logits)
    # the table is really read: another row for the prompt's second token -> other logits
    changed = table.clone()
    changed[int(prompt[0, 1])] += 1.0
    lm2 = model(past_key_values=None, input_ids=prompt, image_features=feats, output_hidden_states=True, token_embeddings=changed)
    assert not torch.equal(lm0.logits, lm2.logits)


def _acc_engine(dev, accum, **kw):
    from grove_amd import train as T
    from grove_amd.synthetic import TINY, synthetic_state_dict
    args = T.shipped_args()
    args.lr, args.grad_accumulation_steps = 1e-3, accum
    model = T.initialize_model(args, dims=TINY, state_dict=synthetic_state_dict(TINY), device=dev)
    eng = T.GroveEngine(model, args, total_steps=1000, **kw)
    eng.scheduler.warm = 0
    return T, args, eng


def test_grad_accumulation_two_micro_steps(dev):
    """train.py:739-782 with --grad_accumulation_steps 2 (DeepSpeed semantics: gradients of the micro-batches are summed, the optimizer
    runs on the boundary with the sum scaled by 1 / steps): the flat gradient after two micro-steps equals g(batch 1) + g(batch 2) of
    two separate one-step engines, engine.step() is a no-op before the boundary, and the update equals the restated AdamW on the mean."""
    from grove_amd.synthetic import TINY, synthetic_batch
    d = TINY
    b1 = to_dev(synthetic_batch(d, B=1, T=8, L=40, n_det=2, seed=31), dev)
    b2 = to_dev(synthetic_batch(d, B=1, T=8, L=44, n_det=3, seed=32), dev)
    T, args, eng = _acc_engine(dev, 2)
    singles = []
    for b in (b1, b2):
        eng.module.zero_grad()
        out = eng.module(**b)
        eng.module.backward(out["loss"])
        torch.cuda.synchronize()
        singles.append(eng.module._flat_grad.clone())
    eng.module.zero_grad()
    w0 = eng.master.clone()
    out = eng(**b1)
    eng.backward(out["loss"])
    eng.step()                      # not a boundary: nothing may move
    torch.cuda.synchronize()
    assert eng.global_step == 0 and torch.equal(eng.master, w0)
    out = eng(**b2)
    eng.backward(out["loss"])
    torch.cuda.synchronize()
    acc = eng.module._flat_grad.clone()
    want = singles[0] + singles[1]
    scale = want.abs().max().item()
    assert (acc - want).abs().max().item() <= 2e-3 * scale, ((acc - want).abs().max().item(), scale)  # (fp32 atomics: split-K / CE sums)
    eng.step()
    torch.cuda.synchronize()
    assert eng.global_step == 1
    gs = acc.double() / 2
    norm = float(gs.pow(2).sum().sqrt())
    gs = gs * (1.0 if norm <= 1.0 else 1.0 / (norm + 1e-6))
    m = (1 - args.beta1) * gs
    v = (1 - args.beta2) * gs * gs
    upd = (m / (1 - args.beta1)) / ((v / (1 - args.beta2)).sqrt() + 1e-8)
    ref = w0.double() - args.lr * upd
    touched = torch.zeros_like(acc, dtype=torch.bool)
    for n, off, k, w in eng.slices:
        touched[off:off + k] = True
    assert (eng.master.double() - ref)[touched].abs().max().item() <= 1e-6 + 1e-3 * args.lr
    assert abs(eng.last_grad_norm - norm) <= 1e-3 * norm


def test_train_mask_decoder_off_trains_the_two_heads_only(dev):
    """train.py:279-288 without --train_mask_decoder: of the decoder only bbox_prediction_head and temporal_objectness_head train. Their
    gradients equal the full policy's (same forward, same backward path through the decoder), the transformer's tensors have no
    gradient slot and do not move in a step."""
    from grove_amd import train as T
    from grove_amd.model.decoder import M_
    from grove_amd.synthetic import TINY, synthetic_batch, synthetic_state_dict
    d = TINY
    batch = to_dev(synthetic_batch(d, B=1, T=8, L=40, n_det=2, seed=5), dev)
    res = {}
    for flag in (True, False):
        args = T.parse_args(["--lora_r", "0", "--pretrained"] + (["--train_mask_decoder"] if flag else []))
        args.lr = 1e-3
        model = T.initialize_model(args, dims=d, state_dict=synthetic_state_dict(d), device=dev)
        assert T.prepare_model_for_training(model, None, args) == model.trainable
        eng = T.GroveEngine(model, args, total_steps=1000)
        eng.scheduler.warm = 0
        before = {k: v.clone() for k, v in model.state_dict().items() if k.startswith(M_)}
        out = eng(**batch)
        eng.backward(out["loss"])
        torch.cuda.synchronize()
        grads = {n: model._grad[n].clone() for n in model.trainable}
        eng.step()
        torch.cuda.synchronize()
        after = {k: v for k, v in model.state_dict().items() if k.startswith(M_)}
        res[flag] = (model.trainable, grads, before, after, float(out["loss"]))
    full, heads = res[True], res[False]
    assert abs(full[4] - heads[4]) <= 1e-5 * abs(full[4])
    dec_heads = [n for n in heads[0] if n.startswith(M_)]
    assert dec_heads and all("bbox_prediction_head" in n or "temporal_objectness_head" in n for n in dec_heads)
    for n in heads[0]:  # two separately built models: equal up to the accumulation-order noise of the fp32 atomics (split-K, CE, scatter-adds)
        a, b = heads[1][n].flatten().double(), full[1][n].flatten().double()
        if float(b.norm()) < 1e-9:
            assert float(a.norm()) < 1e-9, n
            continue
        cos = float(torch.nn.functional.cosine_similarity(a, b, dim=0))
        assert cos > 0.9995 and abs(float(a.norm() / b.norm()) - 1.0) < 5e-3, (n, cos, float(a.norm() / b.norm()))
    moved = [k for k in heads[2] if not torch.equal(heads[2][k], heads[3][k])]
    assert moved and all("bbox_prediction_head" in k or "temporal_objectness_head" in k for k in moved), moved
    assert any("transformer" in k for k in full[2] if not torch.equal(full[2][k], full[3][k]))


def test_vocabulary_not_a_multiple_of_eight(dev):
    """The real vocabulary is 32000 + the added special tokens (train.py:132-152 then resize_token_embeddings, :330) — 32009 with the
    GLaMM-GranD tokenizer: not a multiple of 8. Training step (CE over V columns, lm_head wgrad / dgrad, embedding scatter) and greedy
    decoding on a model with an odd vocabulary, against the oracle."""
    import dataclasses
    from grove_amd import GROVEForCausalLM
    from grove_amd.synthetic import TINY, synthetic_batch, synthetic_state_dict
    from oracle import grove_oracle as O
    d = dataclasses.replace(TINY, vocab=323, det_token_idx=322)
    sd = synthetic_state_dict(d)
    sd_r = {k: v.to(bf).float() for k, v in sd.items()}
    tm = GROVEForCausalLM(dims=d, device=dev, state_dict=sd, det_token_idx=d.det_token_idx, num_frames=8, pe_dtype=torch.float32, train=True)
    tb = synthetic_batch(d, B=2, T=8, L=48, n_det=2, seed=1, ragged=True)
    tm.zero_grad()
    out = tm(**to_dev(tb, dev))
    tm.backward(out["loss"])
    names = ["lm_head.weight", "model.embed_tokens.weight"]
    sdg = {k: v.clone().requires_grad_(k in names) for k, v in sd_r.items()}
    ref = O.model_forward(sdg, d, **oracle_kwargs(tb))
    ref["loss"].backward()
    assert abs(float(out["ce_loss"]) - float(ref["ce_loss"])) <= 2e-2 * abs(float(ref["ce_loss"]))
    for n in names:  # (measured 0.9797 for embed_tokens on this batch: bf16 dx rows; the vocab-320 case of test_model_gpu sits at 0.98-0.99)
        gr, r = tm._grad[n].float().cpu().reshape(-1), sdg[n].grad.reshape(-1)
        assert float(torch.nn.functional.cosine_similarity(gr, r, dim=0)) > 0.97, n
    im = GROVEForCausalLM(dims=d, device=dev, state_dict=sd, det_token_idx=d.det_token_idx, num_frames=8, pe_dtype=torch.float32)
    ib = synthetic_batch(d, B=2, T=8, L=24, n_det=1, seed=3)
    prompt = ib.input_ids[:, :14]
    feats, _ = im(mode="encode_images", images=ib.global_enc_images.to(dev).to(bf))
    g = im.generate(input_ids=prompt.to(dev), image_features=feats, max_new_tokens=6, return_dict_in_generate=True)
    with torch.no_grad():
        of, _ = O.encode_images(sd_r, d, ib.global_enc_images.to(bf).float())
        emb = torch.zeros(16, d.sam_out, d.sam_grid, d.sam_grid)
        oids = O.evaluate(sd_r, d, of, emb, prompt, ib.original_size_list, max_tokens_new=6)[0]
    assert g.sequences.shape[1] <= oids.shape[1] and int(g.sequences.max()) < d.vocab
    n = g.sequences.shape[1]
    assert (g.sequences.cpu()[:, :n] == oids[:, :n]).float().mean().item() > 0.9   # (near-tie logits may flip a token in bf16)


def test_last_layer_tail_equals_the_full_last_layer(dev):
    """Round 4: in a training step only the labelled rows and the [DET] rows of the LLaMA output are read, all in the answer's tail, so
    the last layer runs its queries / o_proj / MLP on positions >= s0 only (keys and values of every position). Same losses and the same
    gradients as the full last layer (`llama_tail=False`) up to accumulation-order noise — both against each other and, through the
    other parity tests, against the oracle."""
    from grove_amd import GROVEForCausalLM
    from grove_amd.model.GROVE import trainable_names
    from grove_amd.synthetic import TINY, synthetic_batch, synthetic_state_dict
    d = TINY
    sd = synthetic_state_dict(d)
    batch = to_dev(synthetic_batch(d, B=2, T=8, L=48, n_det=2, seed=1, ragged=True), dev)
    res = {}
    for tail in (True, False):
        m = GROVEForCausalLM(dims=d, device=dev, state_dict=sd, det_token_idx=d.det_token_idx, num_frames=8, pe_dtype=torch.float32, train=True,
                             llama_tail=tail)
        m.zero_grad()
        out = m(**batch)
        took = isinstance(m._ctx.llama_ctx[0][-1][0], str)
        assert took == tail, "the tail path must be taken exactly when asked (the answer is < half of the 575 + L positions)"
        m.backward(out["loss"])
        torch.cuda.synchronize()
        res[tail] = ({k: float(v) for k, v in out.items() if k.endswith("loss")}, m._flat_grad.clone(), {n: m._grad[n].clone() for n in trainable_names(d)})
    for k in res[False][0]:
        assert abs(res[True][0][k] - res[False][0][k]) <= 2e-3 * max(1.0, abs(res[False][0][k])), (k, res[True][0], res[False][0])
    for n, gf in res[False][2].items():
        gt = res[True][2][n]
        a, b = gt.flatten().double(), gf.flatten().double()
        if float(b.norm()) < 1e-9 or n.endswith("k_proj.bias"):  # (a key bias shifts every score of a row alike: its gradient is pure rounding noise)
            continue
        cos = float(torch.nn.functional.cosine_similarity(a, b, dim=0))
        assert cos > 0.995 and abs(float(a.norm() / b.norm()) - 1.0) < 2e-2, (n, cos, float(a.norm() / b.norm()))
