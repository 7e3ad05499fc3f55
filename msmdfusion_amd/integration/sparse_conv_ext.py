"""`sparse_conv_ext` -- the pybind module of mmdet3d/ops/spconv (src/all.cc:21-51), on
the C ABI: the three functions mmdet3d/ops/spconv/ops.py:48-137 and functional.py:20-75
call for a 3-D SubMConv3d / SparseConv3d.  2-D / 4-D, transposed, pooling, fused-BN and
half variants are outside the hot path and raise.

Formats are the reference's (spconv_ops.h:55-59): `indicePairs` int32 [K,2,N] padded
with -1 ([k][0] input rows, [k][1] output rows), `indiceNum` int32 [K]; strided-conv
output rows come in ascending linear id, the order of the reference's CUDA path
(torch::_unique, spconv_ops.h:119-137).  `filters` is spconv-1.x's
[kd,kh,kw,Cin,Cout] (mmdet3d/ops/spconv/conv.py:100-102).
"""
import torch

from .. import kernels as K

# the output-stationary tables the conv kernels read are the library's native rulebook;
# get_indice_pairs_3d remembers them for the pair tensors it hands out
_TABLES = {}


def _tables_from_pairs(indice_pairs, indice_num, n_in, n_out):
    """(nbr_fwd[K,n_out], nbr_bwd[K,n_in], bwd_is_fwd) of a reference-format rulebook;
    bwd_is_fwd: the second table is the forward one, to be read with the weights
    mirrored in k (SubM rulebooks built here)."""
    hit = _TABLES.get(indice_pairs.data_ptr())
    if hit is not None and hit[0] is indice_pairs and hit[1].shape[1] == n_out:
        return hit[1], hit[2], hit[3]
    kvol, _, ld = indice_pairs.shape
    live = torch.arange(ld, device=indice_pairs.device)[None, :] < indice_num[:, None].long()
    k_of = torch.arange(kvol, device=indice_pairs.device)[:, None].expand(kvol, ld)[live]
    i_of, o_of = indice_pairs[:, 0][live].long(), indice_pairs[:, 1][live].long()
    fwd = torch.full((kvol, n_out), -1, dtype=torch.int32, device=indice_pairs.device)
    bwd = torch.full((kvol, n_in), -1, dtype=torch.int32, device=indice_pairs.device)
    fwd[k_of, o_of] = i_of.int()
    bwd[k_of, i_of] = o_of.int()
    return fwd, bwd, False


def get_indice_pairs_3d(indices, batch, outShape, spatialShape, ksize, stride, padding, dilation,
                        outPadding, subM, transpose):
    """spconv::getIndicePair<3> (spconv_ops.h:33-140) -> [outIds, indicePairs, indiceNum]."""
    if transpose:
        raise RuntimeError("get_indice_pairs_3d: transposed convolutions are not built")
    if any(int(d) != 1 for d in dilation):
        raise RuntimeError("get_indice_pairs_3d: only dilation 1 is built")
    if indices.dim() != 2 or indices.shape[1] != 4 or indices.dtype != torch.int32:
        raise RuntimeError("get_indice_pairs_3d: indices must be int32 [N,4] (b,z,y,x)")
    n = indices.shape[0]
    with torch.cuda.device(indices.device):
        if subM:
            nbr = K.rulebook_subm(indices, batch, spatialShape, ksize)
            pairs, num = K.rulebook_pairs(nbr, ld=max(n, 1))
            _TABLES[pairs.data_ptr()] = (pairs, nbr, nbr, True)
            _trim()
            return [indices, pairs, num]
        out_ids, nbr_fwd, nbr_bwd, out_shape = K.rulebook_conv(indices, batch, spatialShape, ksize,
                                                               stride, padding)
        if [int(x) for x in outShape] != [int(x) for x in out_shape]:
            raise RuntimeError("get_indice_pairs_3d: outShape %s does not match the convolution "
                               "geometry (%s)" % (list(outShape), list(out_shape)))
        pairs, num = K.rulebook_pairs(nbr_fwd, ld=max(n, 1))
        _TABLES[pairs.data_ptr()] = (pairs, nbr_fwd, nbr_bwd, False)
        _trim()
        return [out_ids, pairs, num]


def _trim(keep=64):
    while len(_TABLES) > keep:
        _TABLES.pop(next(iter(_TABLES)))


def _kio(filters):
    if filters.dim() < 3:
        raise RuntimeError("filters must be [k..., Cin, Cout]")
    return filters.reshape(-1, filters.shape[-2], filters.shape[-1])


def indice_conv_fp32(features, filters, indicePairs, indiceNum, numActOut, inverse, subM):
    """spconv::indiceConv<float> (spconv_ops.h:260-361) -> output features [numActOut,Cout]."""
    if inverse:
        raise RuntimeError("indice_conv_fp32: inverse convolutions are not built")
    w = _kio(filters)
    with torch.cuda.device(features.device):
        fwd, _, _ = _tables_from_pairs(indicePairs, indiceNum, features.shape[0], int(numActOut))
        return K.conv_forward(features, K.pack_weight(w), fwd, int(numActOut), w.shape[2])


def indice_conv_backward_fp32(features, filters, outGrad, indicePairs, indiceNum, inverse, subM):
    """spconv::indiceConvBackward<float> (spconv_ops.h:363-456) -> [inputGrad, filtersGrad]."""
    if inverse:
        raise RuntimeError("indice_conv_backward_fp32: inverse convolutions are not built")
    w = _kio(filters)
    n_in = features.shape[0]
    with torch.cuda.device(features.device):
        _, bwd, flip = _tables_from_pairs(indicePairs, indiceNum, n_in, outGrad.shape[0])
        d_in = K.conv_forward(outGrad, K.pack_weight(w, transpose=True), bwd, n_in, w.shape[1],
                              weight_flip=flip)
        d_w = K.conv_wgrad(features, outGrad, indicePairs.contiguous(), indiceNum)
        return [d_in, d_w.view(filters.shape)]


def _not_built(name):
    def fn(*args, **kwargs):
        raise RuntimeError("sparse_conv_ext.%s is outside the MSMDFusion hot path (not built)" % name)
    fn.__name__ = name
    return fn


for _n in ("get_indice_pairs_2d", "get_indice_pairs_4d", "get_indice_pairs_grid_2d",
           "get_indice_pairs_grid_3d", "indice_conv_half", "indice_conv_backward_half",
           "fused_indice_conv_fp32", "fused_indice_conv_half", "indice_maxpool_fp32",
           "indice_maxpool_backward_fp32", "indice_maxpool_half", "indice_maxpool_backward_half"):
    globals()[_n] = _not_built(_n)
