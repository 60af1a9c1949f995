!------------------------------------------------------------------------------
! lart_out -- the output containers of the reference behind one small interface (iofile_mod.f90:39-75 is the
! reference's facade): a file is a sequence of named sections, a section is a table (named f64 columns) or an image
! (N-D f64 array) plus keywords.
!   HDF5 (par%file_format = 'hdf5', the reference's default): group = section EXTNAME, `data` dataset or one dataset
!     per column, keywords = group attributes (schema hdf5io_mod.f90:8-13), Fortran array order.
!   FITS (par%file_format = 'fits'): BinTable / IMAGE HDUs with the same names and keywords, gzip-compressed to
!     <name>.fits.gz like the CFITSIO path of the reference.
!------------------------------------------------------------------------------
module lart_out
  use, intrinsic :: iso_c_binding
  use, intrinsic :: iso_fortran_env, only: real64, int32, int64
  use lart_fits
  implicit none
  private
  integer, parameter :: wp = real64
  integer, parameter, public :: OUT_FITS = 1, OUT_HDF5 = 2
  integer, public :: out_bitpix = 0        ! par%out_bitpix, set by the caller before out_open (-32: float32 storage)

  type, public :: key_t
     character(len=16) :: name = ''
     integer :: kind = 0                    ! 1 real, 2 integer, 3 logical, 4 string
     real(wp) :: r = 0.0_wp
     integer :: i = 0
     logical :: l = .false.
     character(len=64) :: s = ''
     character(len=48) :: comment = ''
  end type key_t
  type, public :: key_list
     integer :: n = 0
     type(key_t) :: k(64)
  end type key_list
  type, public :: out_file
     integer :: fmt = OUT_FITS
     type(fits_file) :: f
     integer(c_long_long) :: h5 = -1
     character(len=512) :: path = ''
  end type out_file

  interface kadd
     module procedure kadd_r, kadd_i, kadd_l, kadd_s
  end interface
  public :: kadd, out_format_of, out_open, out_table, out_image, out_close

  interface
     function lh5_available() bind(C, name='lh5_available') result(a)
       import; integer(c_int) :: a
     end function
     subroutine lh5_set_bitpix(bp) bind(C, name='lh5_set_bitpix')
       import; integer(c_int), value :: bp
     end subroutine
     function lh5_create(path) bind(C, name='lh5_create') result(id)
       import; character(kind=c_char), intent(in) :: path(*); integer(c_long_long) :: id
     end function
     function lh5_group(file, name) bind(C, name='lh5_group') result(id)
       import; integer(c_long_long), value :: file; character(kind=c_char), intent(in) :: name(*); integer(c_long_long) :: id
     end function
     function lh5_put_f64(grp, name, rank, fdims, a) bind(C, name='lh5_put_f64') result(rc)
       import; integer(c_long_long), value :: grp; character(kind=c_char), intent(in) :: name(*); integer(c_int), value :: rank
       integer(c_long_long), intent(in) :: fdims(*); real(c_double), intent(in) :: a(*); integer(c_int) :: rc
     end function
     function lh5_attr_f64(grp, name, v) bind(C, name='lh5_attr_f64') result(rc)
       import; integer(c_long_long), value :: grp; character(kind=c_char), intent(in) :: name(*); real(c_double), value :: v; integer(c_int) :: rc
     end function
     function lh5_attr_i32(grp, name, v) bind(C, name='lh5_attr_i32') result(rc)
       import; integer(c_long_long), value :: grp; character(kind=c_char), intent(in) :: name(*); integer(c_int), value :: v; integer(c_int) :: rc
     end function
     function lh5_attr_str(grp, name, v) bind(C, name='lh5_attr_str') result(rc)
       import; integer(c_long_long), value :: grp; character(kind=c_char), intent(in) :: name(*), v(*); integer(c_int) :: rc
     end function
     function lh5_close_group(grp) bind(C, name='lh5_close_group') result(rc)
       import; integer(c_long_long), value :: grp; integer(c_int) :: rc
     end function
     function lh5_close(file) bind(C, name='lh5_close') result(rc)
       import; integer(c_long_long), value :: file; integer(c_int) :: rc
     end function
     function lart_gzip_file(path) bind(C, name='lart_gzip_file') result(rc)
       import; character(kind=c_char), intent(in) :: path(*); integer(c_int) :: rc
     end function
  end interface

contains
  function cstr(s) result(c)
    character(len=*), intent(in) :: s
    character(kind=c_char, len=:), allocatable :: c
    c = trim(s)//c_null_char
  end function cstr

  subroutine kadd_r(kl, name, v, comment)
    type(key_list), intent(inout) :: kl; character(len=*), intent(in) :: name, comment; real(wp), intent(in) :: v
    kl%n = kl%n + 1; kl%k(kl%n)%name = name; kl%k(kl%n)%kind = 1; kl%k(kl%n)%r = v; kl%k(kl%n)%comment = comment
  end subroutine
  subroutine kadd_i(kl, name, v, comment)
    type(key_list), intent(inout) :: kl; character(len=*), intent(in) :: name, comment; integer, intent(in) :: v
    kl%n = kl%n + 1; kl%k(kl%n)%name = name; kl%k(kl%n)%kind = 2; kl%k(kl%n)%i = v; kl%k(kl%n)%comment = comment
  end subroutine
  subroutine kadd_l(kl, name, v, comment)
    type(key_list), intent(inout) :: kl; character(len=*), intent(in) :: name, comment; logical, intent(in) :: v
    kl%n = kl%n + 1; kl%k(kl%n)%name = name; kl%k(kl%n)%kind = 3; kl%k(kl%n)%l = v; kl%k(kl%n)%comment = comment
  end subroutine
  subroutine kadd_s(kl, name, v, comment)
    type(key_list), intent(inout) :: kl; character(len=*), intent(in) :: name, v, comment
    kl%n = kl%n + 1; kl%k(kl%n)%name = name; kl%k(kl%n)%kind = 4; kl%k(kl%n)%s = v; kl%k(kl%n)%comment = comment
  end subroutine

  !--- 'hdf5' / 'h5' -> HDF5 when the library is in this build, anything else (or no library) -> FITS
  function out_format_of(file_format) result(fmt)
    character(len=*), intent(in) :: file_format
    integer :: fmt
    fmt = OUT_FITS
    if ((trim(file_format) == 'hdf5' .or. trim(file_format) == 'h5') .and. lh5_available() /= 0) fmt = OUT_HDF5
  end function out_format_of

  subroutine out_open(o, base, fmt)
    type(out_file), intent(out) :: o
    character(len=*), intent(in) :: base
    integer, intent(in) :: fmt
    o%fmt = fmt
    if (fmt == OUT_HDF5) then
       o%path = trim(base)//'.h5'
       call lh5_set_bitpix(int(out_bitpix, c_int))
       o%h5 = lh5_create(cstr(o%path))
       if (o%h5 < 0) then
          write(*,'(2a)') 'ERROR: cannot create ', trim(o%path); stop 1
       endif
    else
       o%path = trim(base)//'.fits'
       fits_bitpix = merge(-32, -64, out_bitpix == -32)
       call fits_open(o%f, trim(o%path))
    endif
  end subroutine out_open

  subroutine put_keys_fits(f, kl)
    type(fits_file), intent(inout) :: f
    type(key_list), intent(in) :: kl
    integer :: j
    do j = 1, kl%n
       select case (kl%k(j)%kind)
       case (1); call fits_key(f, trim(kl%k(j)%name), kl%k(j)%r, trim(kl%k(j)%comment))
       case (2); call fits_key(f, trim(kl%k(j)%name), kl%k(j)%i, trim(kl%k(j)%comment))
       case (3); call fits_key(f, trim(kl%k(j)%name), kl%k(j)%l, trim(kl%k(j)%comment))
       case (4); call fits_key(f, trim(kl%k(j)%name), trim(kl%k(j)%s), trim(kl%k(j)%comment))
       end select
    enddo
  end subroutine put_keys_fits

  subroutine put_keys_h5(g, kl)
    integer(c_long_long), intent(in) :: g
    type(key_list), intent(in) :: kl
    integer :: j, rc
    do j = 1, kl%n
       select case (kl%k(j)%kind)
       case (1); rc = lh5_attr_f64(g, cstr(kl%k(j)%name), kl%k(j)%r)
       case (2); rc = lh5_attr_i32(g, cstr(kl%k(j)%name), int(kl%k(j)%i, c_int))
       case (3); rc = lh5_attr_i32(g, cstr(kl%k(j)%name), merge(1_c_int, 0_c_int, kl%k(j)%l))      ! hdf5_put_key_logical: integer 0/1
       case (4); rc = lh5_attr_str(g, cstr(kl%k(j)%name), cstr(kl%k(j)%s))
       end select
    enddo
  end subroutine put_keys_h5

  !--- table section: f64 columns cols(nrow, ncol)
  subroutine out_table(o, extname, names, cols, kl)
    type(out_file), intent(inout) :: o
    character(len=*), intent(in) :: extname, names(:)
    real(wp), intent(in) :: cols(:,:)
    type(key_list), intent(in) :: kl
    integer(c_long_long) :: g, d(1)
    integer :: j, rc
    if (o%fmt == OUT_HDF5) then
       g = lh5_group(o%h5, cstr(extname)); d(1) = size(cols, 1)
       do j = 1, size(names)
          rc = lh5_put_f64(g, cstr(names(j)), 1_c_int, d, cols(:, j))
       enddo
       call put_keys_h5(g, kl); rc = lh5_attr_str(g, cstr('EXTNAME'), cstr(extname)); rc = lh5_close_group(g)
    else
       call fits_begin_table(o%f, size(cols, 1), size(names), names)
       call put_keys_fits(o%f, kl)
       call fits_key(o%f, 'EXTNAME', extname, extname)
       call fits_end_header(o%f)
       call fits_write_table_rows(o%f, cols)
    endif
  end subroutine out_table

  !--- image section: f64 array of Fortran extents naxes(:)
  subroutine out_image(o, extname, naxes, a, kl)
    type(out_file), intent(inout) :: o
    character(len=*), intent(in) :: extname
    integer, intent(in) :: naxes(:)
    real(wp), intent(in) :: a(*)
    type(key_list), intent(in) :: kl
    integer(c_long_long) :: g, d(8)
    integer :: rc
    if (o%fmt == OUT_HDF5) then
       g = lh5_group(o%h5, cstr(extname)); d(1:size(naxes)) = naxes
       rc = lh5_put_f64(g, cstr('data'), int(size(naxes), c_int), d, a)
       call put_keys_h5(g, kl); rc = lh5_attr_str(g, cstr('EXTNAME'), cstr(extname)); rc = lh5_close_group(g)
    else
       call fits_begin_image(o%f, naxes)
       call fits_key(o%f, 'EXTNAME', extname, extname)
       call put_keys_fits(o%f, kl)
       call fits_end_header(o%f)
       o%f%nbytes = product(int(naxes, int64))
       call fits_write_image(o%f, a)
    endif
  end subroutine out_image

  !--- close; FITS files are gzip-compressed to <name>.fits.gz when gz is set (the reference's form)
  subroutine out_close(o, gz)
    type(out_file), intent(inout) :: o
    logical, intent(in) :: gz
    integer :: rc
    if (o%fmt == OUT_HDF5) then
       rc = lh5_close(o%h5)
    else
       call fits_close(o%f)
       if (gz) then
          rc = lart_gzip_file(cstr(o%path))
          if (rc == 0) o%path = trim(o%path)//'.gz'
       endif
    endif
  end subroutine out_close
end module lart_out
