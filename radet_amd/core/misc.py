from functools import partial


def multi_apply(func, *args, **kwargs):
    """radet/core/utils/misc.py: apply func to each tuple of args and transpose the results."""
    pfunc = partial(func, **kwargs) if kwargs else func
    return tuple(map(list, zip(*map(pfunc, *args))))
