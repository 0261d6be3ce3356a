"""
Host-side pieces of round 6 that need no GPU: the split-K rule of the latency plan (layers/conv.latency_split).
"""
from keras_retinanet_3D.layers import conv as C

# (name, K, C_in, C_out, output pixels per image at 402 x 1333)
LAYERS = [('res3 2b', 3, 128, 128, 8517), ('res4 2a', 1, 1024, 256, 2184), ('res4 2b', 3, 256, 256, 2184), ('res4 2c', 1, 256, 1024, 2184),
          ('res5 2a', 1, 2048, 512, 546), ('res5 2b', 3, 512, 512, 546), ('res5 2c', 1, 512, 2048, 546), ('C5_reduced', 1, 2048, 512, 546),
          ('P5', 3, 512, 512, 546), ('P6', 3, 2048, 512, 147), ('P7', 3, 512, 512, 44), ('C4_reduced', 1, 1024, 512, 2184), ('P4', 3, 512, 512, 2184),
          ('P3', 3, 512, 512, 8517), ('towers 0', 3, 512, 896, 11438), ('reg 1', 3, 512, 512, 11438), ('cls 1', 3, 256, 256, 11438)]


def test_the_default_rule_is_the_librarys():
    """ csrc/conv_igemm.hip split_rule, restated on the host: only res5 branch2b and P5 ... P7 split in the default plan """
    got = {n: C.default_split(k, k, ci, co, pix) for n, k, ci, co, pix in LAYERS}
    assert {n for n, s in got.items() if s > 1} == {'res5 2b', 'P5', 'P6', 'P7'}
    assert got['res5 2b'] == 3 and got['P6'] == 8


def test_the_latency_rule_is_a_function_of_the_layer_alone_and_never_below_the_default():
    got = {n: C.latency_split(k, k, ci, co, pix) for n, k, ci, co, pix in LAYERS}
    for n, k, ci, co, pix in LAYERS:
        assert got[n] >= C.default_split(k, k, ci, co, pix) and 1 <= got[n] <= 8
        assert got[n] == C.latency_split(k, k, ci, co, pix)                 # no hidden state: the batch size is not even an argument
    # the deep-K layers whose batch-1 grid leaves most CUs idle split; the big grids and the shallow layers do not
    assert got['res4 2b'] > 1 and got['res5 2a'] > 1 and got['P4'] > 1 and got['C5_reduced'] > 1 and got['res5 2b'] > 3
    assert got['res3 2b'] == got['res4 2c'] == got['P3'] == got['towers 0'] == got['reg 1'] == got['cls 1'] == 1
    # every split keeps at least 16 K-steps (512 values of K) per part
    for n, k, ci, co, pix in LAYERS:
        if got[n] > C.default_split(k, k, ci, co, pix):
            assert k * k * ci // got[n] >= 512
