// Internal (library-private, C++ linkage) entry points of sparse_conv_st.hip, called by the dispatcher in sparse_conv.hip.
#pragma once

// window rows of the SUPER-TILE form of a shape (0 = the shape has none): the tile book a caller passes must be built for it
int pvx_st_rcap(int cin, int cout, int kvol, int dtype);

// PVX_OK when the launch was made, PVX_ERR_RANGE when this form does not take the call (caller falls through to the other forms)
int pvx_st_launch(const void* feat, int n_in, int cin, const void* prepared_weight, const float* bias, const int* nbr, int n_out,
                  const int* n_out_dev, int cout, int dtype, const float* scale, const float* shift, const void* residual, int relu,
                  void* out, const void* tilebook, int tilebook_rcap, void* stream);

// deals the tiles of `book` to the workgroups of the kernel that reads it, by cost (tilebook_layout.h, SCHEDULE): 1 = a schedule was
// launched, 0 = that kernel takes none, -1 = launch error
int pvx_st_schedule(void* book, int n_out, const int* n_out_dev, int cin, int cout, int kvol, int dtype, void* stream);
