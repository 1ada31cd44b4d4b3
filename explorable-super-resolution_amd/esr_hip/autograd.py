"""Backward passes (data-gradient / weight-gradient / CEM adjoints).  Filled in after the forward path."""


def _todo(what):
    raise NotImplementedError('%s: the HIP backward kernels are not built yet; run under torch.no_grad() '
                              '(there is deliberately no stock-PyTorch fallback)' % what)


def rrdb_forward_with_grad(engine, x, pad):
    _todo('RRDBNet backward')


def conv3x3_function(x, weight, bias, act_slope, split):
    _todo('conv3x3 backward')


def cem_project_with_grad(*a, **k):
    _todo('CEM backward')


class CemLinear:
    @staticmethod
    def apply(*a, **k):
        _todo('CEM filter backward')
