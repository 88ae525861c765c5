"""Single-query serving path — host-side mirror of dvl/utils.py:204-211 (retrieve_query) and of the bulk encoder
dvl/utils.py:214-233 (get_model_encoded_vecs), on the MI355X indexer.  The towers stay PyTorch-ROCm host code; the
query vector never leaves the device on its way into the search (the reference's ``.detach().cpu().numpy()`` at
:210 is gone)."""
import torch


def retrieve_query(model, query, indexer, args, top=10):
    """dvl/utils.py:204-211 — tokenise one string, run the text tower, search the index for the 100 best rows.
    (Like the reference, ``top`` is accepted but the search depth is fixed at 100.)"""
    input_ids = args.tokenizer.encode(query)
    input_ids = torch.LongTensor(input_ids).to(args.device).unsqueeze(0)
    attn_mask = torch.ones(len(input_ids[0]), dtype=torch.long, device=args.device).unsqueeze(0)
    pos_ids = torch.arange(len(input_ids[0]), dtype=torch.long, device=args.device).unsqueeze(0)
    with torch.no_grad():
        _, query_vector, _ = model.txt_model(input_ids=input_ids, attention_mask=attn_mask, position_ids=pos_ids)
    return indexer.search_knn(query_vector.detach(), 100)


def pool_cls(sequence_output: torch.Tensor, normalize: bool = False, out_bf16: bool = False):
    """[CLS] pooling kernel (dvl/models/bi_encoder.py:120,188: ``sequence_output[:, 0, :]``), optionally fused with
    the opt-in L2 normalisation and the bf16 cast used by the index ingest.  Returns fp32 [B, D] (and bf16 [B, D])."""
    import ctypes
    from . import _lib as L
    lib = L.load_library()
    if not sequence_output.is_cuda:
        raise L.LdotError(-2, 'pool_cls needs a CUDA(HIP) tensor: there is no CPU fallback')
    s = sequence_output.detach()
    if s.stride(-1) != 1 or s.stride(1) != s.shape[2]:
        s = s.contiguous()
    B, Ls, D = s.shape
    code = {torch.float32: L.F32, torch.bfloat16: L.BF16, torch.float16: L.F16}[s.dtype]
    o32 = torch.empty((B, D), dtype=torch.float32, device=s.device)
    o16 = torch.empty((B, D), dtype=torch.bfloat16, device=s.device) if out_bf16 else None
    L.check(lib.ldot_cls_pool(ctypes.c_void_p(s.data_ptr()), code, B, s.stride(0), D, int(normalize),
                              ctypes.c_void_p(o32.data_ptr()),
                              ctypes.c_void_p(o16.data_ptr()) if o16 is not None else ctypes.c_void_p(0),
                              ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)))
    return (o32, o16) if out_bf16 else o32
