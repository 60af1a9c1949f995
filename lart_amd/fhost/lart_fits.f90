!------------------------------------------------------------------------------
! Minimal FITS writer for the output sections of write_output_rect.f90
! (:423-521 `Spectrum` binary table + keywords; :1113-1350 peel-off image HDUs).
! Uncompressed standard FITS (the reference writes the same HDUs through CFITSIO and gzip):
! 2880-byte blocks, 80-character cards, big-endian IEEE data.  No external library.
!------------------------------------------------------------------------------
module lart_fits
  use, intrinsic :: iso_fortran_env, only: real32, real64, int32, int64
  implicit none
  private
  public :: fits_bitpix
  public :: fits_file, fits_open, fits_close, fits_begin_table, fits_begin_image, fits_key, fits_end_header, &
            fits_write_table_rows, fits_write_image

  type fits_file
     integer :: unit = -1
     integer(int64) :: nbytes = 0          ! bytes written in the current header / data unit
  end type fits_file

  integer :: fits_bitpix = -64             ! par%out_bitpix: -32 writes float32 images and 1E table columns

  interface fits_key
     module procedure key_r8, key_i4, key_l, key_s
  end interface

contains
  subroutine put_card(f, card)
    type(fits_file), intent(inout) :: f
    character(len=*), intent(in) :: card
    character(len=80) :: c
    c = card
    write(f%unit) c
    f%nbytes = f%nbytes + 80
  end subroutine put_card

  subroutine pad_block(f, fill)
    type(fits_file), intent(inout) :: f
    character(len=1), intent(in) :: fill
    integer :: npad, i
    npad = int(mod(2880_int64 - mod(f%nbytes, 2880_int64), 2880_int64))
    do i = 1, npad
       write(f%unit) fill
    enddo
    f%nbytes = 0
  end subroutine pad_block

  subroutine fits_open(f, filename)
    type(fits_file), intent(out) :: f
    character(len=*), intent(in) :: filename
    open(newunit=f%unit, file=filename, access='stream', form='unformatted', status='replace', convert='big_endian')
    f%nbytes = 0
    call put_card(f, 'SIMPLE  =                    T / file conforms to FITS standard')
    call put_card(f, 'BITPIX  =                    8')
    call put_card(f, 'NAXIS   =                    0')
    call put_card(f, 'EXTEND  =                    T')
    call put_card(f, 'END')
    call pad_block(f, ' ')
  end subroutine fits_open

  subroutine fits_close(f)
    type(fits_file), intent(inout) :: f
    close(f%unit); f%unit = -1
  end subroutine fits_close

  subroutine key_r8(f, name, v, comment)
    type(fits_file), intent(inout) :: f
    character(len=*), intent(in) :: name, comment
    real(real64), intent(in) :: v
    character(len=80) :: c
    character(len=8) :: k8
    k8 = name                              ! keyword names are left-justified in their 8 columns
    write(c, '(a8,"= ",es20.12e3," / ",a)') k8, v, comment(1:min(len(comment), 47))        ! a card holds 80 characters
    call put_card(f, c)
  end subroutine key_r8
  subroutine key_i4(f, name, v, comment)
    type(fits_file), intent(inout) :: f
    character(len=*), intent(in) :: name, comment
    integer, intent(in) :: v
    character(len=80) :: c
    character(len=8) :: k8
    k8 = name                              ! keyword names are left-justified in their 8 columns
    write(c, '(a8,"= ",i20," / ",a)') k8, v, comment(1:min(len(comment), 47))
    call put_card(f, c)
  end subroutine key_i4
  subroutine key_l(f, name, v, comment)
    type(fits_file), intent(inout) :: f
    character(len=*), intent(in) :: name, comment
    logical, intent(in) :: v
    character(len=80) :: c
    character(len=8) :: k8
    k8 = name                              ! keyword names are left-justified in their 8 columns
    write(c, '(a8,"= ",a20," / ",a)') k8, merge('                   T', '                   F', v), comment(1:min(len(comment), 47))
    call put_card(f, c)
  end subroutine key_l
  subroutine key_s(f, name, v, comment)
    type(fits_file), intent(inout) :: f
    character(len=*), intent(in) :: name, v, comment
    character(len=80) :: c
    character(len=8) :: k8
    character(len=18) :: q
    k8 = name                              ! keyword names are left-justified in their 8 columns
    q = v
    write(c, '(a8,"= ''",a,"'' / ",a)') k8, q(1:max(8, len_trim(v))), comment(1:min(len(comment), 47))
    call put_card(f, c)
  end subroutine key_s

  !--- binary table of ncol float64 columns, nrow rows (io_append_table_column with bitpix=-64)
  subroutine fits_begin_table(f, nrow, ncol, names)
    type(fits_file), intent(inout) :: f
    integer, intent(in) :: nrow, ncol
    character(len=*), intent(in) :: names(ncol)
    character(len=8) :: kw
    integer :: i
    f%nbytes = 0
    call put_card(f, "XTENSION= 'BINTABLE'           / binary table extension")
    call key_i4(f, 'BITPIX', 8, 'array data type')
    call key_i4(f, 'NAXIS', 2, 'number of array dimensions')
    call key_i4(f, 'NAXIS1', merge(4, 8, fits_bitpix == -32)*ncol, 'length of dimension 1')
    call key_i4(f, 'NAXIS2', nrow, 'length of dimension 2')
    call key_i4(f, 'PCOUNT', 0, 'number of group parameters')
    call key_i4(f, 'GCOUNT', 1, 'number of groups')
    call key_i4(f, 'TFIELDS', ncol, 'number of table fields')
    do i = 1, ncol
       write(kw, '("TTYPE",i0)') i; call key_s(f, kw, trim(names(i)), 'label for field')
       write(kw, '("TFORM",i0)') i
       if (fits_bitpix == -32) then
          call key_s(f, kw, '1E', 'data format of field: 4-byte REAL')
       else
          call key_s(f, kw, '1D', 'data format of field: 8-byte DOUBLE')
       endif
    enddo
  end subroutine fits_begin_table

  subroutine fits_begin_image(f, naxes)
    type(fits_file), intent(inout) :: f
    integer, intent(in) :: naxes(:)
    character(len=8) :: kw
    integer :: i
    f%nbytes = 0
    call put_card(f, "XTENSION= 'IMAGE   '           / IMAGE extension")
    call key_i4(f, 'BITPIX', fits_bitpix, 'number of bits per data pixel')
    call key_i4(f, 'NAXIS', size(naxes), 'number of data axes')
    do i = 1, size(naxes)
       write(kw, '("NAXIS",i0)') i; call key_i4(f, kw, naxes(i), 'length of data axis')
    enddo
    call key_i4(f, 'PCOUNT', 0, 'required keyword; must = 0')
    call key_i4(f, 'GCOUNT', 1, 'required keyword; must = 1')
  end subroutine fits_begin_image

  subroutine fits_end_header(f)
    type(fits_file), intent(inout) :: f
    call put_card(f, 'END')
    call pad_block(f, ' ')
  end subroutine fits_end_header

  subroutine fits_write_table_rows(f, cols)
    type(fits_file), intent(inout) :: f
    real(real64), intent(in) :: cols(:,:)        ! (nrow, ncol)
    integer :: i, j
    do i = 1, size(cols, 1)
       do j = 1, size(cols, 2)
          if (fits_bitpix == -32) then
             write(f%unit) real(cols(i, j), real32)
          else
             write(f%unit) cols(i, j)
          endif
       enddo
    enddo
    f%nbytes = merge(4_int64, 8_int64, fits_bitpix == -32)*size(cols)
    call pad_block(f, achar(0))
  end subroutine fits_write_table_rows

  subroutine fits_write_image(f, a)
    type(fits_file), intent(inout) :: f
    real(real64), intent(in) :: a(*)
    integer(int64) :: n
    n = f%nbytes                                     ! caller passes the element count through nbytes
    if (fits_bitpix == -32) then
       write(f%unit) real(a(1:n), real32)
    else
       write(f%unit) a(1:n)
    endif
    f%nbytes = merge(4_int64, 8_int64, fits_bitpix == -32)*n
    call pad_block(f, achar(0))
  end subroutine fits_write_image
end module lart_fits
