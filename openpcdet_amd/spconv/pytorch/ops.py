"""spconv.pytorch.ops helpers the reference's tools call
(libraries/3DSparseConvolution/tool/centerpoint-export/exptool.py:76-82)."""


def get_conv_output_size(input_size, kernel_size, stride, padding, dilation):
    out = []
    for i in range(len(input_size)):
        size = (input_size[i] + 2 * padding[i] - dilation[i] * (kernel_size[i] - 1) - 1) // stride[i] + 1
        out.append(1 if kernel_size[i] == -1 else size)
    return out


def get_deconv_output_size(input_size, kernel_size, stride, padding, dilation, output_padding):
    out = []
    for i in range(len(input_size)):
        if kernel_size[i] == -1:
            raise ValueError("deconv don't support kernel_size < 0")
        out.append((input_size[i] - 1) * stride[i] - 2 * padding[i] + kernel_size[i] + output_padding[i])
    return out
