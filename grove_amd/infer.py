"""Sliding-window clip inference: the host logic of infer_iground.py:110-148 (frame sampler) and
:150-288 (`inference`), on top of GROVEForCausalLM's modes.

The reference captions the CENTRE window with `evaluate` and then runs every other window one at a time through
`model_forward(inference=True)` with the generated answer appended to the prompt (teacher forced), keeping only the frames a
window sees first. Here all remaining windows of a clip go through ONE batched forward (windows are independent samples of
the batch dimension), and the caption / dense PE / prompt are reused; per-frame outputs come back in frame order.

Text handling: the reference decodes the generated ids, strips markup and re-tokenises "prompt + answer" through its
conversation template (infer_iground.py:62-85, 243-244). No tokenizer ships offline, so the default here feeds the generated
ids back as they are (pads dropped, sequence cut after eos); pass `answer_ids_fn(generated_row) -> 1-D ids` to plug in the
tokenizer round trip.
"""
import torch


def sliding_segment_with_mask(num_frames=48, num_segments=8):
    """infer_iground.py:110-148. Window j samples frame j of each of `num_segments` equal segments (stride = segment size);
    the `num_frames % num_segments` left-over frames get extra windows shifted past the last full offset. masks[j][k] = 1
    iff window j is the first to see its k-th frame."""
    seg, rem = divmod(num_frames, num_segments)
    offsets = list(range(seg)) + [seg + r for r in range(rem)]
    all_indices, masks, seen = [], [], set()
    for off in offsets:
        idx = [i * seg + off for i in range(num_segments)]
        if off >= seg:
            idx = [f for f in idx if f < num_frames]
            if not idx:
                continue
        masks.append([0 if f in seen else 1 for f in idx])
        all_indices.append(idx)
        seen.update(idx)
    return all_indices, masks


def centre_window(masks):
    """Index of the window whose caption is generated (infer_iground.py:166-170): the middle of the all-new windows."""
    last = 0
    for i, m in enumerate(masks):
        if all(m):
            last = i
    return last // 2


def default_answer_ids(row, pad_token_id, eos_token_id):
    """Generated row -> ids to teacher-force in the other windows: pads dropped, cut after the first eos."""
    keep = row[row != pad_token_id]
    eos = (keep == eos_token_id).nonzero().flatten()
    if eos.numel():
        keep = keep[:int(eos[0]) + 1]
    return keep


@torch.no_grad()
def infer_clip(model, global_enc_images_all, grounding_enc_images_all, prompt_ids, original_size, *, max_tokens_new=64,
               answer_ids_fn=None, token_embeddings=None, num_segments=8):
    """One clip: global_enc_images_all [1, 3, F, 336, 336], grounding_enc_images_all [1, 3, F, 512, 512] (F >= num_segments),
    prompt_ids 1-D (one -200). Returns a dict with per-FRAME lists in frame order: `pred_bboxes` (xyxy pixels, thresholded),
    `logits_temp_objectness`, `frame_indices`; plus `output_ids` (the centre window's generated row) and `windows`."""
    dev = model.dev
    F = global_enc_images_all.shape[2]
    all_indices, masks = sliding_segment_with_mask(F, num_segments)
    c = centre_window(masks)
    sizes = [original_size]

    def pick(x, idx):
        return x[:, :, idx].contiguous()

    feats, outs = model(mode="encode_images", images=pick(global_enc_images_all, all_indices[c]))
    emb = model(mode="get_grounding_encoder_embs", images=pick(grounding_enc_images_all, all_indices[c]))
    ids, boxes, logits = model(mode="evaluate", image_features=feats, image_forward_outs=outs, images_dtype=global_enc_images_all.dtype,
                               image_embeddings=emb, input_ids=prompt_ids[None].to(dev), original_size_list=sizes,
                               max_tokens_new=max_tokens_new, token_embeddings=token_embeddings)
    per_frame = {}
    for k, f in enumerate(all_indices[c]):
        per_frame[f] = (boxes[0][k], logits[0][k])
    d = model.dims
    row = ids[0].cpu()
    answer = answer_ids_fn(row) if answer_ids_fn is not None else default_answer_ids(row, d.pad_token_id, d.eos_token_id)
    rest = [j for j in range(len(all_indices)) if j != c and len(all_indices[j]) == num_segments]
    short = [j for j in range(len(all_indices)) if j != c and len(all_indices[j]) != num_segments]
    if short:
        raise ValueError("a trailing window with fewer than num_segments frames cannot run: the model reshapes T=8 groups "
                         "(the reference fails on it as well)")
    if rest:
        W = len(rest)
        g = torch.cat([pick(global_enc_images_all, all_indices[j]) for j in rest], 0)
        s = torch.cat([pick(grounding_enc_images_all, all_indices[j]) for j in rest], 0)
        ids_w = answer[None].repeat(W, 1).to(dev)
        preds = model(global_enc_images=g, grounding_enc_images=s, bboxes_region=None, input_ids=ids_w, labels=None,
                      attention_masks=None, offset=None, bboxes_list=None, temp_objectness_labels_list=None,
                      original_size_list=sizes * W, inference=True)
        for w, j in enumerate(rest):
            for k, f in enumerate(all_indices[j]):
                if masks[j][k]:
                    per_frame[f] = (preds["pred_bboxes"][w][k], preds["logits_temp_objectness"][w][k])
    frames = sorted(per_frame)
    return {"frame_indices": frames, "pred_bboxes": [per_frame[f][0] for f in frames],
            "logits_temp_objectness": [per_frame[f][1] for f in frames], "output_ids": row, "answer_ids": answer,
            "windows": all_indices, "centre": c}


def update_and_sort_video_outputs(gathered_results):
    """infer_iground.py:87-108: merge the per-rank result dicts in rank order; the first occurrence of a clip id wins (the
    DistributedSampler pads the last round by wrap-around, so a clip can come back from two ranks)."""
    video_outputs = {}
    for process_results in gathered_results:
        for clip_id, data in process_results.items():
            if clip_id not in video_outputs:
                video_outputs[clip_id] = data
    return video_outputs


def _to_host(x):
    """Tensors of a result leave the device before they are pickled for the gather (the reference stores host lists too)."""
    if torch.is_tensor(x):
        return x.detach().cpu()
    if isinstance(x, (list, tuple)):
        return type(x)(_to_host(v) for v in x)
    if isinstance(x, dict):
        return {k: _to_host(v) for k, v in x.items()}
    return x


@torch.no_grad()
def infer_dataset(model, clips, prompt_ids, *, rank=None, world=None, max_tokens_new=64, answer_ids_fn=None, token_embeddings=None,
                  num_segments=8, gather=True, on_clip=None):
    """The multi-rank inference job of infer_iground.py:150-293, 538-551: `clips` is an indexable dataset whose item i is
    (clip_id, global_enc_images_all [1, 3, F, 336, 336], grounding_enc_images_all [1, 3, F, 512, 512], original_size) — or a callable
    `clips.load(i)` style object with `__len__` / `__getitem__`; every rank takes the clip indices `shard_clips(len(clips), rank, world)`
    (the un-shuffled DistributedSampler partition, wrap-around padded), runs `infer_clip` on each, then — exactly as the reference —
    `barrier` + `all_gather_object` of the per-rank {clip_id: result} dicts and the first-wins merge on every rank. Replicas only:
    there is no data-path collective, the gather moves the (host) results once at the end. Without an initialised process group
    (or world == 1) it is the plain loop. Returns the merged dict (every rank holds it; the reference pickles rank 0's)."""
    import torch.distributed as dist
    from .train import shard_clips
    if world is None:
        world = dist.get_world_size() if dist.is_initialized() else 1
    if rank is None:
        rank = dist.get_rank() if dist.is_initialized() else 0
    mine = {}
    for i in shard_clips(len(clips), rank, world):
        clip_id, g_all, s_all, size = clips[i]
        if clip_id in mine:  # wrap-around padding handed this rank a clip twice
            continue
        res = infer_clip(model, g_all.to(model.dev), s_all.to(model.dev), prompt_ids, size, max_tokens_new=max_tokens_new,
                         answer_ids_fn=answer_ids_fn, token_embeddings=token_embeddings, num_segments=num_segments)
        mine[clip_id] = _to_host(res)
        if on_clip is not None:
            on_clip(clip_id, mine[clip_id])
    if world > 1 and gather:
        dist.barrier()                                     # infer_iground.py:290
        parts = [None for _ in range(world)]
        dist.all_gather_object(parts, mine)                # :291-292
        return update_and_sort_video_outputs(parts)        # :293
    return mine
