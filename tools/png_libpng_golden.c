/* Provenance of tests/golden/*_gray_libpng.png: decode a PNG with REAL libpng 1.6.37 the way
 * hpc/read_img.c does (png_set_rgb_to_gray(png, 1, -1, -1)) and dump the gray bytes.
 * Build container only: gcc png_libpng_golden.c -I/opt/conda/include -L/opt/conda/lib -lpng16 -lz */
#include <png.h>
#include <stdio.h>
#include <stdlib.h>
int main(int argc,char**argv){FILE*f=fopen(argv[1],"rb");png_structp p=png_create_read_struct(PNG_LIBPNG_VER_STRING,0,0,0);png_infop i=png_create_info_struct(p);png_init_io(p,f);png_read_info(p,i);int w=png_get_image_width(p,i),h=png_get_image_height(p,i);int ct=png_get_color_type(p,i);if(ct==PNG_COLOR_TYPE_RGB||ct==PNG_COLOR_TYPE_RGB_ALPHA)png_set_rgb_to_gray(p,1,-1,-1);png_read_update_info(p,i);int rb=png_get_rowbytes(p,i);png_bytep*rows=malloc(sizeof(png_bytep)*h);for(int y=0;y<h;y++)rows[y]=malloc(rb);png_read_image(p,rows);fprintf(stderr,"%d %d ct=%d rb=%d\n",w,h,ct,rb);FILE*o=fopen(argv[2],"wb");for(int y=0;y<h;y++)fwrite(rows[y],1,w*(rb/w),o);fclose(o);return 0;}
