"""`get_conv_paddings` for string paddings ("same"/"valid"); no arithmetic on tensors."""


def get_conv_paddings(kernel_size, stride, padding, dilation):
    if padding == "valid":
        return 0, 0
    if padding == "same":
        if stride != 1:
            raise ValueError("'same' padding requires stride 1")
        total = dilation * (kernel_size - 1)
        left = total // 2
        return left, total - left
    raise ValueError(f"unknown padding {padding!r}")
