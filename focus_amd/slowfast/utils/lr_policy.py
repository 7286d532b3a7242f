"""Step schedules of the slot training loop: the temperature anneal and the learning-rate warm-up that
steve_train_net.py:60-80 evaluates every iteration (reference: slowfast/utils/lr_policy.py:8-40, repeated in
slowfast/models/STEVE/utils.py:8-44).  The epoch-based policies of that file belong to the supervised loop's
scheduler, which stays torch-side.

Both schedules are a clamp outside [start_step, final_step) around a shape function of the elapsed fraction; the
arithmetic inside keeps the reference's operation order so the values agree to the last bit."""
import math


def _clamped(step, start_value, final_value, start_step, final_step, inside):
    if start_step > final_step:
        raise AssertionError("schedule window is reversed: %r > %r" % (start_step, final_step))
    if step < start_step:
        return start_value
    if step >= final_step:
        return final_value
    return inside(final_step - start_step)


def cosine_anneal(step, start_value, final_value, start_step, final_step):
    """Half a cosine period from start_value down to final_value."""
    if start_value < final_value:
        raise AssertionError("cosine_anneal only decays")
    half_span, middle = 0.5 * (start_value - final_value), 0.5 * (start_value + final_value)
    return _clamped(step, start_value, final_value, start_step, final_step,
                    lambda width: half_span * math.cos(math.pi * ((step - start_step) / width)) + middle)


def linear_warmup(step, start_value, final_value, start_step, final_step):
    """Straight line up to final_value; the elapsed fraction counts the current step as done (step + 1)."""
    if start_value > final_value:
        raise AssertionError("linear_warmup only grows")
    return _clamped(step, start_value, final_value, start_step, final_step,
                    lambda width: (final_value - start_value) * ((step + 1 - start_step) / width) + start_value)
